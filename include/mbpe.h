/*
 * mbpe.h -- C-ABI of the MI355X-native BPE trainer hot path.
 *
 * Drop-in boundary for justinhj/minbpe-cc's training path (both tie-breaks) and its encode.
 * The reference has no FFI layer of its own; every entry point
 * below names the reference code it replaces (paths relative to the
 * reference checkout, code/include/...).  INTEGRATION.md shows the binding a
 * reference maintainer would add inside Tokenizer::train.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross the boundary
 *   - every function returns 0 (MBPE_OK) or a negative mbpe_status; the text
 *     of the last error of the calling thread is mbpe_last_error()
 *   - the caller owns every buffer it passes; the context owns device memory
 *   - one context per GPU and per thread (thread-compatible, like the
 *     reference's Tokenizer, which is not thread-safe: Tokenizer.h:67-72)
 *   - there is NO CPU fallback: without a usable HIP device mbpe_create fails
 *
 * Token ids: the reference's Token is a uint32_t (Tokenizer.h:37-38).  The device stream holds
 * 16-bit slots while the ids allow it: up to MBPE_MAX_VOCAB_BASIC for a single-chunk corpus and
 * MBPE_MAX_VOCAB_CHUNKED when chunk boundaries are present (up to MBPE_MAX_VOCAB_ENDBIT a chunked
 * stream marks "last token of its chunk" with one slot bit; beyond that it keeps a barrier slot
 * after every chunk instead: one more slot per chunk, ids use all 16 bits).  A larger vocab_size
 * (up to MBPE_MAX_VOCAB_WIDE) trains its first merges on the slot stream and the rest on 32-bit
 * tokens with 64-bit pair keys (csrc/wide.h: one merge per pass; lexical tie-break, one GPU --
 * with the `first` tie-break or several ranks such a request still returns MBPE_ERR_VOCAB).
 */
#ifndef MBPE_H
#define MBPE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBPE_API __attribute__((visibility("default")))

#define MBPE_MAX_VOCAB_BASIC   65534u
#define MBPE_MAX_VOCAB_ENDBIT  32766u
#define MBPE_MAX_VOCAB_CHUNKED 65518u
#define MBPE_MAX_VOCAB_WIDE    16777216u   /* 2^24: the 32-bit continuation */
#define MBPE_NO_BARRIER 0xFFFFFFFFu

typedef enum {
    MBPE_NEED_EXCHANGE =  1,  /* external-transport mode only: reduce the exchange buffer, then
                                 call mbpe_comm_exchange_done */
    MBPE_OK            =  0,
    MBPE_ERR_ARG       = -1,  /* bad argument (NULL, vocab_size < 256: Tokenizer.h:492) */
    MBPE_ERR_NO_DEVICE = -2,  /* no HIP device / extension unusable */
    MBPE_ERR_HIP       = -3,  /* a HIP runtime call failed */
    MBPE_ERR_VOCAB     = -4,  /* vocab_size beyond MBPE_MAX_VOCAB_WIDE, or beyond the 16-bit slot format where the
                                 32-bit continuation does not apply (`first` tie-break, several ranks) */
    MBPE_ERR_STATE     = -5,  /* call order violated (e.g. steps before begin) */
    MBPE_ERR_OOM       = -6,  /* device or host allocation failed */
    MBPE_ERR_REGEX     = -7,  /* PCRE2 unavailable, compile or match error */
    MBPE_ERR_SPLIT_GAP = -8,  /* (unused since gaps between matches are skipped like the reference does) */
    MBPE_ERR_COMM      = -9,  /* RCCL unavailable or a collective failed */
    MBPE_ERR_OVERFLOW  = -10, /* pair table or count overflow detected on device */
    MBPE_ERR_IO        = -11  /* file could not be read / written */
} mbpe_status;

typedef struct mbpe_ctx mbpe_ctx;

/* Per-run statistics; all times are milliseconds of device time measured
 * with HIP events on the context's stream. */
typedef struct {
    uint64_t n_bytes;          /* corpus bytes loaded */
    uint64_t n_chunks;         /* chunks after dropping NUL-quirk chunks */
    uint64_t n_slots;          /* physical stream slots right now */
    uint64_t n_live;           /* live tokens right now */
    uint32_t n_merges;         /* merges performed so far */
    uint32_t n_compactions;    /* stream compactions performed */
    uint64_t n_pairs;          /* pairs ever inserted (== PairCount::get_count) */
    float    ms_pair_count;    /* last pair-count scan kernel */
    float    ms_begin;         /* widen + table build + first argmax */
    float    ms_steps;         /* all merge steps so far */
    uint32_t pair_count_launches;
    uint32_t merge_launches;   /* merge-kernel launches timed ("time_kernels" option) */
    float    ms_merge_kernel;  /* summed duration of those launches */
    uint32_t n_batches;        /* stream passes that merged something (several merges can share one) */
    uint32_t n_fused;          /* of them: fused passes (large batches, merged stream written to the other buffer) */
    uint32_t n_fused_dropped;  /* fused passes whose output was abandoned because validation kept only a prefix */
    uint32_t cut_conflict;     /* batches ended by a pair that depends on an earlier pair of the batch */
    uint32_t cut_bucket;       /* ... by a full lookup bucket */
    uint32_t cut_single;       /* ... by a (t,t) or zero-count pair (merged alone) */
    uint32_t cut_full;         /* batches that reached the size limit */
    uint32_t n_validation_drops; /* pairs selected but not merged in that pass (validation) */
    float    ms_grow_table;    /* host wall time spent growing the pair table (allocation + rehash) */
    float    ms_compact;       /* host wall time spent compacting the stream */
    uint32_t n_table_grows;
    uint32_t n_sel_fallback;   /* batches chosen by the bound-walking selection instead of the threshold gather */
    uint32_t fused_launches;   /* fused passes timed ("time_kernels" option) */
    float    ms_fused_kernel;  /* their summed duration */
    uint64_t fused_slots;      /* stream slots those passes read (and wrote) */
    uint32_t n_sel_retry;      /* batches whose first candidate gather overflowed (threshold found among the block bounds) */
    uint32_t adapt_limit;      /* current batch size limit learnt from validation */
    uint64_t n_sel_blocks;     /* 1024-entry blocks of the pair table read by the candidate gathers */
    uint32_t size_hist[8];     /* passes by merges committed: 1, 2-3, 4-7, 8-15, 16-31, 32-63, 64-127, 128 and more */
    uint32_t n_skipped;        /* dependent candidates passed over by the selection (merged in a later pass) */
    uint32_t n_skip_cut;       /* ... that had not fallen behind the batch after all (the batch was cut there) */
    uint64_t exchange_words;   /* multi-GPU: u32 words sum-all-reduced for the count deltas of all sequences so far */
    uint32_t exchanges;        /* ... in this many all-reduces (one per sequence) */
    uint32_t pad_;
    uint64_t fused_live_tokens; /* "time_kernels": live tokens before + live tokens after, summed over the timed fused
                                   passes (x 2 bytes = SURVEY 8(d)'s 2 B x L read + 2 B x L' written of those passes) */
    float    ms_pair_count_kernel; /* mbpe_pair_count_u8 without a table: mean KERNEL duration of the call's launches
                                   (start/stop events of each dispatch: no gap between launches, no marker overhead) */
    uint32_t pad2_;
} mbpe_stats;

MBPE_API const char *mbpe_last_error(void);
MBPE_API const char *mbpe_version(void);

/* ---- context ------------------------------------------------------- */

/* Creates a context on HIP device `device_id`.  Fails with
 * MBPE_ERR_NO_DEVICE when there is no such device. */
MBPE_API int  mbpe_create(int device_id, mbpe_ctx **out);
MBPE_API void mbpe_destroy(mbpe_ctx *ctx);

/* ---- corpus -------------------------------------------------------- */

/* Hands the training text to the device.  Replaces text_to_vector +
 * create_lists (Tokenizer.h:85-100, :114-124): the corpus stays a byte
 * array in HBM instead of one heap node per token.
 *   text         n_bytes bytes; host memory, or device memory when
 *                text_on_device != 0 (then it is used in place and must
 *                stay valid until mbpe_train_begin returns)
 *   chunk_off    n_chunks+1 ascending byte offsets, chunk c =
 *                [chunk_off[c], chunk_off[c+1]); chunk_off[0] == 0 and
 *                chunk_off[n_chunks] == n_bytes.  NULL = one chunk = the
 *                whole text (Tokenizer.h:541-544).  Always host memory.
 * Pairs are only counted and merged inside a chunk (Tokenizer.h:135-144,
 * :311-319).  A chunk that starts with NUL and whose remainder parses with
 * std::stoi collapses to one token in the reference (Tokenizer.h:86-93) and
 * therefore never contributes a pair: such chunks are dropped here. */
MBPE_API int mbpe_load_corpus(mbpe_ctx *ctx, const uint8_t *text, uint64_t n_bytes,
                              const uint64_t *chunk_off, uint64_t n_chunks,
                              int text_on_device);

/* The same for chunks given as ranges [starts[c], ends[c]) (ascending, not overlapping) that need
 * not tile the text: bytes outside every chunk are not part of the corpus, as in the reference's
 * match loop (Tokenizer.h:506-540).  When there are such bytes the library trains on a packed copy
 * of the chunks (host memory, or a device text that is copied back once); mbpe_stats.n_bytes is the
 * packed size. */
MBPE_API int mbpe_load_corpus_ranges(mbpe_ctx *ctx, const uint8_t *text, uint64_t n_bytes,
                                     const uint64_t *starts, const uint64_t *ends, uint64_t n_chunks,
                                     int text_on_device);

/* The pair-count scan on the loaded byte corpus: calculate_freqs
 * (Tokenizer.h:127-146) with PairCountLexicalOrder::create_or_modify_pair
 * (PairCount.h:249-260) as one histogram kernel.  table65536_out (host,
 * optional) receives count[(first << 8) | second].  Test / bench
 * granularity; mbpe_train_begin runs the same kernel. */
MBPE_API int mbpe_pair_count_u8(mbpe_ctx *ctx, uint32_t *table65536_out);

/* ---- training ------------------------------------------------------ */

/* Prepares the training loop for `vocab_size` (>= 256, Tokenizer.h:492):
 * pair-count scan, 16-bit slot stream, pair table, first argmax.
 * Corresponds to Tokenizer.h:551-556. */
MBPE_API int mbpe_train_begin(mbpe_ctx *ctx, uint32_t vocab_size);

/* Runs up to n_steps iterations of the loop body Tokenizer.h:557-589:
 * get_top_pair_count (PairCount.h:262-269), merge_chunks ->
 * merge_incremental (Tokenizer.h:309-320, :202-306).  Stops early when the
 * target vocab size is reached or the pair table is empty (:586-588).
 * steps_done_out (optional) receives the number of merges made by this
 * call. */
MBPE_API int mbpe_train_steps(mbpe_ctx *ctx, uint32_t n_steps, uint32_t *steps_done_out);

/* Runs up to n_sequences *batch sequences*, the unit the library executes: select the next
 * maxima that are provably independent (get_top_pair_count, PairCount.h:262-269, for up to
 * "max_batch" consecutive iterations) -> one pass over the token stream that merges them all
 * (merge_chunks, Tokenizer.h:309-320) -> validate against the one-at-a-time order -> apply the
 * count updates (Tokenizer.h:239-280).  Every sequence commits at least one merge of the loop
 * Tokenizer.h:557-589 unless training is complete; merges_done_out (optional) receives the
 * number this call committed.  With "multi_merge" 0 a sequence is one merge. */
MBPE_API int mbpe_train_sequences(mbpe_ctx *ctx, uint32_t n_sequences, uint32_t *merges_done_out);

/* Copies the merges made so far: merges_out[2k], merges_out[2k+1] is the
 * pair that became token 256+k (Tokenizer.h:578); counts_out[k] (optional)
 * is its count when chosen (the verbose line, Tokenizer.h:566-576).
 * cap_merges = capacity of the arrays in merges. */
MBPE_API int mbpe_train_result(mbpe_ctx *ctx, uint32_t *merges_out, int32_t *counts_out,
                               uint32_t cap_merges, uint32_t *n_merges_out);

/* One call = Tokenizer::train's hot path (Tokenizer.h:551-589) for
 * CONFLICT_RESOLUTION::LEXICAL: load + begin + all steps + result.
 * merges_out must hold 2*(vocab_size-256) u32. */
MBPE_API int mbpe_train_lexical(mbpe_ctx *ctx, const uint8_t *text, uint64_t n_bytes,
                                const uint64_t *chunk_off, uint64_t n_chunks,
                                uint32_t vocab_size,
                                uint32_t *merges_out, int32_t *counts_out,
                                uint32_t *n_merges_out, mbpe_stats *stats_out);

/* The same for either CONFLICT_RESOLUTION of Tokenizer::train (minbpe-cc.cpp:129-131):
 * conflict_resolution 1 = LEXICAL (what mbpe_train_lexical runs), 0 = FIRST -- the reference
 * CLI's default: among the pairs of maximal count the one inserted first into a table rebuilt
 * before every merge wins (PairCountInsertOrder, PairCount.h:65-74, :141-152; recount
 * Tokenizer.h:581-585), i.e. the one whose first occurrence in the corpus comes first.  The
 * device keeps the exact counts incrementally and settles ties with one pass over the stream
 * that finds the earliest position of a tied pair; one merge per pass.  On a sharded stream (several ranks) every rank
 * finds the earliest tied pair of its shard and one more small exchange per merge -- the ranks' hits, the lowest rank
 * that has one wins: shards follow each other in rank order -- makes the choice global.
 * With FIRST the loop ends when no pair is left (Tokenizer.h:586-588): n_merges_out may then be
 * smaller than vocab_size - 256.  Step-level use: mbpe_set_option("conflict_resolution", 0)
 * before mbpe_train_begin. */
MBPE_API int mbpe_train(mbpe_ctx *ctx, const uint8_t *text, uint64_t n_bytes,
                        const uint64_t *chunk_off, uint64_t n_chunks, uint32_t vocab_size,
                        int conflict_resolution,
                        uint32_t *merges_out, int32_t *counts_out,
                        uint32_t *n_merges_out, mbpe_stats *stats_out);

MBPE_API int mbpe_get_stats(mbpe_ctx *ctx, mbpe_stats *out);

/* ---- introspection (parity tests) ----------------------------------- */

/* Live tokens of the stream in order (holes removed).  tokens_out may be
 * NULL to query the length.  chunk_end_out (optional, same length) is 1
 * where a token is the last of its chunk. */
MBPE_API int mbpe_get_stream(mbpe_ctx *ctx, uint32_t *tokens_out, uint8_t *chunk_end_out,
                             uint64_t cap, uint64_t *n_out);

/* Device view of the slot stream, for checks that run on the device (tests, bench.py): n_slots
 * slots of slot_bits bits each in device memory, valid until the next training call.  A slot equal
 * to the all-ones value is a hole; end_bit (0 when there is none) is the slot bit that marks the
 * last token of a chunk, the token id is the slot without it; barrier (MBPE_NO_BARRIER when there
 * is none) is the slot value that stands after the last token of every chunk and is no token. */
MBPE_API int mbpe_stream_device(mbpe_ctx *ctx, const void **slots_out, uint64_t *n_slots_out,
                                uint32_t *slot_bits_out, uint32_t *end_bit_out, uint32_t *barrier_out);

/* Device view of the dense pair table (vocab_size <= 32,768 unless "dense_table" is 0; MBPE_ERR_STATE
 * for the hashed layout): 1 << (2 * vshift) u32 cells, cell of (a, b) at
 *   ((((a >> 5) << (vshift - 5)) | (b >> 5)) << 10) | ((a & 31) << 5) | (b & 31),
 * value 0x80000000 | count once the pair was ever inserted (PairCount.h:249-260 never erases), else 0. */
MBPE_API int mbpe_table_device(mbpe_ctx *ctx, const void **cells_out, uint32_t *vshift_out);

/* All pairs ever inserted with their current counts
 * (PairCount::get_all, PairCount.h:271-278; order unspecified).
 * Arrays may be NULL to query the size. */
MBPE_API int mbpe_get_pairs(mbpe_ctx *ctx, uint32_t *first_out, uint32_t *second_out,
                            int32_t *count_out, uint64_t cap, uint64_t *n_out);

/* Forces a stream compaction now (normally triggered by the hole ratio). */
MBPE_API int mbpe_compact(mbpe_ctx *ctx);

/* Tuning knobs (tests force rare paths with them).
 *   "compact_den"   compact when holes * den >= slots (default 16; 0 = never)
 *   "batch"         sequences (or single merges) per host round trip (default 16)
 *   "multi_merge"   1 = several independent merges per stream pass (default), 0 = one
 *   "max_batch"     most merges one pass may take (default and limit 4096; 1024 when the stream
 *                   is sharded over several GPUs, whose exchange grows with it)
 *   "byte_table"    1 = a batch whose pairs are all pairs of raw bytes is looked up in a byte x byte
 *                   table by the stream kernels (default), 0 = always the hashed batch table
 *   "fused_min"     batches of at least this many pairs read the stream once and write
 *                   the merged stream to the second buffer (default 24; frequent pairs
 *                   qualify earlier); 2 = every multi-pair batch, >= 1000 = never
 *   "dense_table"   -1/1 one cell per possible pair when vocab <= 32,768 (default),
 *                   0 = always the hashed pair table
 *   "threshold_select" 1 = choose batches from a gathered, sorted candidate list
 *                   (default), 0 = always walk the argmax bounds pair by pair
 *   "sel_cap"       capacity of the candidate list of the threshold selection (default and
 *                   limit 8192, at least 64; tests lower it to force the overflow path)
 *   "pc_repeat"     mbpe_pair_count_u8 called without an output table launches the scan this many times
 *                   back to back and reports the mean duration in mbpe_stats.ms_pair_count (timing only)
 *   "hier_argmax"   -1 auto / 0 scan every entry / 1 walk the block bounds
 *                   (single-merge mode)
 *   "force_exchange" 1 = take the multi-rank path (rank edges, exchange) even
 *                   with a single rank (tests the RCCL binding on one GPU)
 *   "chunk_barrier" chunk ends of a chunked corpus as barrier slots: -1 (default) when vocab_size
 *                   exceeds MBPE_MAX_VOCAB_ENDBIT, 1 always, 0 never; read by mbpe_train_begin
 *   "first_batches" `first` tie-break only: 1 = pairs whose count no other candidate shares are merged in batches
 *                   like in lexical mode, a pair with a shared count goes alone after the position tie-break; the
 *                   run hands over to the one-merge-per-pass loop when most sequences are such single pairs.
 *                   Default 0 (one merge per pass): same results, and no faster on text.
 *   "conflict_resolution" 1 = lexical tie-break (default), 0 = first (see mbpe_train); before
 *                   mbpe_train_begin only
 *   "time_kernels"  1 = bracket every merge kernel with HIP events on the
 *                   context's stream; totals appear in mbpe_stats
 *   "lockstep"      how the host enqueues batch sequences: 1 = it waits for every selection and enqueues only the
 *                   kernels that sequence needs (one event wait per sequence, ~12 launches instead of ~27), 0 = it
 *                   enqueues whole groups of sequences with every kernel variant and the device decides which work;
 *                   -1 (default): 1 for streams of up to 32 Mi slots on one rank, 0 otherwise.  Same results.
 *   "pair_cells"    a match whose two neighbours are raw bytes costs the stream pass one atomic on a byte x byte cell block
 *                   of its pair (65,536 u32 per pair of the largest batch: 1 GiB at the default "max_batch"), folded into
 *                   the pair's delta rows right behind the pass, instead of two atomics on the rows: 1 / 0, -1 (default) =
 *                   for streams of 64 Mi slots and more, where the passes of thousands of byte pairs are bound by their
 *                   atomics; read by mbpe_train_begin.  Same results.
 *   "wide_from"     tests: hand over to the 32-bit continuation after this many merges whatever the vocabulary
 *                   (-1, the default: where the 16-bit slot format ends); read by mbpe_train_begin
 */
MBPE_API int mbpe_set_option(mbpe_ctx *ctx, const char *name, int64_t value);

/* ---- encode on the device ----------------------------------------------- */

/* internal_encode (Tokenizer.h:370-377) over all chunks of a text on HIP device `device_id`:
 * every chunk is widened with text_to_vector (Tokenizer.h:85-100, including its rule that a chunk
 * starting with NUL whose remainder parses with std::stoi is ONE token with that id -- how encode
 * hands over special tokens, :635-637, :667-670) and run through internal_internal_encode
 * (:325-367): left-to-right passes that replace ANY pair present in merges_lookup, until a pass
 * replaces nothing.  The results are concatenated (:713-717).
 *   text, chunk_off, n_chunks   as for mbpe_load_corpus (host memory; NULL chunk_off = one chunk)
 *   merges                      2 * n_merges u32, merge k makes token 256 + k; a repeated pair keeps
 *                               the last id (merges_lookup[pair] = idx, Tokenizer.h:579)
 *   tokens_out                  may be NULL to query the count; cap = its capacity in tokens
 *   n_passes_out                optional: stream passes made (the deepest chunk's passes)
 * No CPU fallback: MBPE_ERR_NO_DEVICE without a HIP device.  Token ids must stay below 2^31 - 2. */
MBPE_API int mbpe_encode_chunks(int device_id, const uint8_t *text, uint64_t n_bytes,
                                const uint64_t *chunk_off, uint64_t n_chunks,
                                const uint32_t *merges, uint32_t n_merges,
                                uint32_t *tokens_out, uint64_t cap, uint64_t *n_out,
                                uint32_t *n_passes_out);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) ---------------- */

/* Rank 0 creates the id, every rank receives it out of band (the launcher
 * broadcasts it, e.g. over torch.distributed) and calls mbpe_comm_init.
 * After that mbpe_load_corpus takes the rank's contiguous shard of the
 * corpus (whole chunks for chunked corpora; any byte range of a one-chunk
 * corpus) and training all-reduces the per-merge count deltas and boundary
 * descriptors so every rank takes identical decisions. */
#define MBPE_COMM_ID_BYTES 128
MBPE_API int mbpe_comm_unique_id(uint8_t id_out[MBPE_COMM_ID_BYTES]);
MBPE_API int mbpe_comm_init(mbpe_ctx *ctx, const uint8_t id[MBPE_COMM_ID_BYTES],
                            int rank, int n_ranks);

/* The same sharded algorithm with the collective left to the caller (any
 * transport that can sum u32 buffers across ranks: MPI, gloo, a test harness).
 * After mbpe_comm_init_external, mbpe_train_begin and mbpe_train_steps return
 * MBPE_NEED_EXCHANGE each time the ranks have to exchange data: the caller
 * sum-all-reduces the buffer named by mbpe_comm_exchange_buffer (device
 * memory, u32 elements, identical length on every rank) in place and calls
 * mbpe_comm_exchange_done, which continues the operation and returns
 * MBPE_NEED_EXCHANGE again (next merge) or MBPE_OK (operation complete). */
MBPE_API int mbpe_comm_init_external(mbpe_ctx *ctx, int rank, int n_ranks);
MBPE_API int mbpe_comm_exchange_buffer(mbpe_ctx *ctx, void **dev_ptr_out, uint64_t *n_u32_out);
MBPE_API int mbpe_comm_exchange_done(mbpe_ctx *ctx);

/* ---- host-side pieces of the reference path ------------------------- */

/* Regex pre-split, Tokenizer.h:500-540: successive non-empty PCRE2 matches
 * (options PCRE2_UTF|PCRE2_UCP, +PCRE2_CASELESS when the pattern contains
 * "(?i:", :407-415; PCRE2_NO_UTF_CHECK at match time, :512) become chunks
 * [starts[c], ends[c]).  Bytes between matches belong to no chunk: the reference skips them
 * (:506-540; the built-in gpt2/gpt4 patterns leave none on valid UTF-8).  When the chunks tile
 * the text, mbpe_split_offsets gives the n_chunks + 1 offsets mbpe_load_corpus takes; otherwise it
 * returns NULL and the chunks go to mbpe_load_corpus_ranges. */
typedef struct mbpe_split mbpe_split;
MBPE_API int  mbpe_presplit(const char *pattern, const uint8_t *text, uint64_t n_bytes,
                            mbpe_split **out);
MBPE_API uint64_t        mbpe_split_count(const mbpe_split *s);
MBPE_API const uint64_t *mbpe_split_offsets(const mbpe_split *s);
MBPE_API int             mbpe_split_has_gaps(const mbpe_split *s);
MBPE_API const uint64_t *mbpe_split_starts(const mbpe_split *s);
MBPE_API const uint64_t *mbpe_split_ends(const mbpe_split *s);
MBPE_API void            mbpe_split_free(mbpe_split *s);

/* The split patterns of Tokenizer.h:59-60 ("gpt2", "gpt4"; "basic" = ""). */
MBPE_API const char *mbpe_split_pattern(const char *encoder_name);

#ifdef __cplusplus
}
#endif
#endif /* MBPE_H */
