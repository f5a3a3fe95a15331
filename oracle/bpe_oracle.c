/*
 * bpe_oracle.c -- CPU restatement of the minbpe-cc BPE training hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (minbpe-cc_amd/,
 * include/) may link, import or call this file.  It is used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker /
 * the timed CPU baseline ("port"), never as the thing shipped.
 *
 * Parity pin: the reference cannot be built in this image (Boost.MultiIndex,
 * pcre2.h, CLI11 and <expected> are absent and writing stand-ins for them is
 * not allowed), so this restatement is pinned by
 *   - the reference's own known-answer tests, code/test/test.cpp:82-106
 *     (lexical tie-break) and :136-186 ("abcbcde" count/merge KAT),
 *   - the sha256 digests of reference-produced .model files recorded in
 *     SURVEY.md section 8c (taylorswift/shakespeare/SplitMix64/small/aaaa),
 * both checked in tests/test_oracle.py.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to the reference checkout).
 *
 * Plain C11, single-threaded like the reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* pair -> count table with never-erased membership                    */
/* restates PairCountLexicalOrder / PairCountInsertOrder,              */
/* code/include/PairCount.h:101-181 and :227-279                       */
/* ------------------------------------------------------------------ */

typedef struct {
    uint64_t key;       /* (first << 32) | second */
    int32_t  count;
    uint8_t  used;
    uint64_t order;     /* insertion order (InsertOrder variant, PairCount.h:149) */
} pc_slot;

typedef struct {
    uint64_t key;
    int32_t  count;
} heap_ent;

typedef struct {
    pc_slot *slots;
    uint64_t cap;       /* power of two */
    uint64_t n;         /* number of pairs ever inserted == get_count(), PairCount.h:235 */
    uint64_t next_order;
    /* ordered index 1 (PairCount.h:216-217) restated as a lazy max-heap:
     * every count change pushes (count,key); stale tops are dropped on read. */
    heap_ent *heap;
    uint64_t hn, hcap;
    int use_heap;
} pc_table;

static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static void pc_init(pc_table *t, int use_heap) {
    memset(t, 0, sizeof(*t));
    t->cap = 1024;
    t->slots = (pc_slot *)calloc(t->cap, sizeof(pc_slot));
    t->use_heap = use_heap;
}

static void pc_free(pc_table *t) {
    free(t->slots);
    free(t->heap);
    memset(t, 0, sizeof(*t));
}

static pc_slot *pc_find(pc_table *t, uint64_t key) {
    uint64_t m = t->cap - 1, i = mix64(key) & m;
    while (t->slots[i].used) {
        if (t->slots[i].key == key) return &t->slots[i];
        i = (i + 1) & m;
    }
    return NULL;
}

static void pc_grow(pc_table *t) {
    pc_slot *old = t->slots;
    uint64_t ocap = t->cap;
    t->cap *= 2;
    t->slots = (pc_slot *)calloc(t->cap, sizeof(pc_slot));
    uint64_t m = t->cap - 1;
    for (uint64_t j = 0; j < ocap; j++) {
        if (!old[j].used) continue;
        uint64_t i = mix64(old[j].key) & m;
        while (t->slots[i].used) i = (i + 1) & m;
        t->slots[i] = old[j];
    }
    free(old);
}

/* heap order: CompareLexicalOrder, PairCount.h:194-207
 * (count descending, then first ascending, then second ascending);
 * key packs (first,second) so key ascending == lexical ascending. */
static int heap_before(const heap_ent *a, const heap_ent *b) {
    if (a->count != b->count) return a->count > b->count;
    return a->key < b->key;
}

static void heap_push(pc_table *t, uint64_t key, int32_t count) {
    if (t->hn == t->hcap) {
        t->hcap = t->hcap ? t->hcap * 2 : 1024;
        t->heap = (heap_ent *)realloc(t->heap, t->hcap * sizeof(heap_ent));
    }
    uint64_t i = t->hn++;
    heap_ent e = {key, count};
    while (i > 0) {
        uint64_t p = (i - 1) / 2;
        if (!heap_before(&e, &t->heap[p])) break;
        t->heap[i] = t->heap[p];
        i = p;
    }
    t->heap[i] = e;
}

static void heap_pop(pc_table *t) {
    heap_ent e = t->heap[--t->hn];
    uint64_t i = 0;
    for (;;) {
        uint64_t c = 2 * i + 1;
        if (c >= t->hn) break;
        if (c + 1 < t->hn && heap_before(&t->heap[c + 1], &t->heap[c])) c++;
        if (!heap_before(&t->heap[c], &e)) break;
        t->heap[i] = t->heap[c];
        i = c;
    }
    if (t->hn) t->heap[i] = e;
}

/* create_or_modify_pair, PairCount.h:249-260 (and :141-152): find; if
 * present count += freq, else insert with count = freq. Returns 1 if new. */
static int pc_add(pc_table *t, uint32_t a, uint32_t b, int32_t freq, int defer_heap) {
    uint64_t key = ((uint64_t)a << 32) | b;
    pc_slot *s = pc_find(t, key);
    if (s) {
        s->count += freq;
        if (t->use_heap && !defer_heap) heap_push(t, key, s->count);
        return 0;
    }
    if ((t->n + 1) * 2 > t->cap) pc_grow(t);
    uint64_t m = t->cap - 1, i = mix64(key) & m;
    while (t->slots[i].used) i = (i + 1) & m;
    t->slots[i].used = 1;
    t->slots[i].key = key;
    t->slots[i].count = freq;
    t->slots[i].order = t->next_order++;
    t->n++;
    if (t->use_heap && !defer_heap) heap_push(t, key, freq);
    return 1;
}

/* get_pair, PairCount.h:239-247: present? -> count */
static int pc_get(pc_table *t, uint32_t a, uint32_t b, int32_t *count) {
    pc_slot *s = pc_find(t, ((uint64_t)a << 32) | b);
    if (!s) return 0;
    if (count) *count = s->count;
    return 1;
}

static void pc_build_heap(pc_table *t) {
    for (uint64_t j = 0; j < t->cap; j++)
        if (t->slots[j].used) heap_push(t, t->slots[j].key, t->slots[j].count);
}

/* get_top_pair_count, lexical: PairCount.h:262-269 = begin() of the index
 * ordered by CompareLexicalOrder over ALL pairs ever inserted (never erased,
 * so zero-count pairs stay candidates: SURVEY.md 8-S rule 4). */
static int pc_top_lexical(pc_table *t, uint32_t *a, uint32_t *b, int32_t *count) {
    while (t->hn) {
        heap_ent *e = &t->heap[0];
        pc_slot *s = pc_find(t, e->key);
        if (s && s->count == e->count) {
            *a = (uint32_t)(e->key >> 32);
            *b = (uint32_t)e->key;
            if (count) *count = e->count;
            return 1;
        }
        heap_pop(t);
    }
    return 0;
}

/* get_top_pair_count, insertion order: PairCount.h:159-166 with
 * CompareCountOrder :66-74 (count descending, then insert_order ascending). */
static int pc_top_first(pc_table *t, uint32_t *a, uint32_t *b, int32_t *count) {
    pc_slot *best = NULL;
    for (uint64_t j = 0; j < t->cap; j++) {
        pc_slot *s = &t->slots[j];
        if (!s->used) continue;
        if (!best || s->count > best->count ||
            (s->count == best->count && s->order < best->order))
            best = s;
    }
    if (!best) return 0;
    *a = (uint32_t)(best->key >> 32);
    *b = (uint32_t)best->key;
    if (count) *count = best->count;
    return 1;
}

/* ------------------------------------------------------------------ */
/* trainer state: chunks of tokens (reference: vector<forward_list>)   */
/* ------------------------------------------------------------------ */

typedef struct {
    uint32_t *tok;       /* all chunks back to back */
    uint64_t *off;       /* n_chunks + 1 ; chunk c = tok[off[c] .. off[c] + len[c]) */
    uint64_t *len;       /* live length of each chunk */
    uint64_t  n_chunks;
    pc_table  tab;
    int       mode;      /* 0 = lexical, 1 = first */
} orc_state;

/* text_to_vector, code/include/Tokenizer.h:85-100, including the NUL quirk
 * :86-93: a chunk starting with '\0' whose remainder parses with std::stoi
 * becomes ONE token with that value.  Returns the number of tokens written. */
static int stoi_prefix(const uint8_t *s, uint64_t n, long long *out) {
    /* std::stoi == strtol base 10: skip isspace, optional sign, >=1 digit,
     * value must fit in int (else out_of_range -> caught -> fallback). */
    uint64_t i = 0;
    while (i < n && (s[i] == ' ' || (s[i] >= 9 && s[i] <= 13))) i++;
    int neg = 0;
    if (i < n && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    if (i >= n || s[i] < '0' || s[i] > '9') return 0;
    long long v = 0;
    while (i < n && s[i] >= '0' && s[i] <= '9') {
        v = v * 10 + (s[i] - '0');
        if (v > 4294967296LL) return 0; /* far outside int: out_of_range */
        i++;
    }
    if (neg) v = -v;
    if (v > 2147483647LL || v < -2147483648LL) return 0;
    *out = v;
    return 1;
}

static uint64_t text_to_vector(const uint8_t *s, uint64_t n, uint32_t *out) {
    if (n > 0 && s[0] == 0) {
        long long id;
        if (stoi_prefix(s + 1, n - 1, &id)) {
            out[0] = (uint32_t)(int)id;
            return 1;
        }
    }
    for (uint64_t i = 0; i < n; i++) out[i] = s[i]; /* char_to_token :80-82 */
    return n;
}

/* calculate_freqs (forward_list overload), Tokenizer.h:127-146 */
static void calculate_freqs(orc_state *st) {
    pc_free(&st->tab);
    pc_init(&st->tab, st->mode == 0);
    for (uint64_t c = 0; c < st->n_chunks; c++) {
        const uint32_t *t = st->tok + st->off[c];
        for (uint64_t i = 0; i + 1 < st->len[c]; i++)
            pc_add(&st->tab, t[i], t[i + 1], 1, 1);
    }
    if (st->mode == 0) pc_build_heap(&st->tab);
}

ORC_EXPORT void orc_destroy(orc_state *st) {
    if (!st) return;
    free(st->tok); free(st->off); free(st->len);
    pc_free(&st->tab);
    free(st);
}

/* Builds the trainer state the way Tokenizer::train does before its loop,
 * Tokenizer.h:498-552: chunks (chunk_off == NULL -> one chunk = whole text,
 * :541-544), create_lists :114-124, calculate_freqs :127-146. */
ORC_EXPORT orc_state *orc_create(const uint8_t *text, uint64_t n_bytes,
                                 const uint64_t *chunk_off, uint64_t n_chunks,
                                 int mode) {
    orc_state *st = (orc_state *)calloc(1, sizeof(orc_state));
    uint64_t one[2] = {0, n_bytes};
    if (!chunk_off) { chunk_off = one; n_chunks = 1; }
    st->mode = mode;
    st->n_chunks = n_chunks;
    st->tok = (uint32_t *)malloc((n_bytes + 1) * sizeof(uint32_t));
    st->off = (uint64_t *)malloc((n_chunks + 1) * sizeof(uint64_t));
    st->len = (uint64_t *)malloc((n_chunks + 1) * sizeof(uint64_t));
    for (uint64_t c = 0; c < n_chunks; c++) {
        st->off[c] = chunk_off[c];
        st->len[c] = text_to_vector(text + chunk_off[c], chunk_off[c + 1] - chunk_off[c],
                                    st->tok + chunk_off[c]);
    }
    st->off[n_chunks] = n_bytes;
    pc_init(&st->tab, mode == 0);
    calculate_freqs(st);
    return st;
}

/* get_top_pair_count through the PairCount interface, Tokenizer.h:558 */
ORC_EXPORT int orc_top(orc_state *st, uint32_t *a, uint32_t *b, int32_t *count) {
    return st->mode == 0 ? pc_top_lexical(&st->tab, a, b, count)
                         : pc_top_first(&st->tab, a, b, count);
}

/* merge_incremental, Tokenizer.h:202-306, on one chunk.  The reference walks
 * a forward_list with cursors i0,i1,i2; here r is i1, r+1 is i2 and the
 * already-written output t[w-1] is i0 (the possibly rewritten left neighbour).
 * Update order and the "only decrement if present" guards follow :239-280. */
static uint64_t merge_incremental(uint32_t *t, uint64_t n, uint32_t a, uint32_t b,
                                  uint32_t x, pc_table *tab) {
    uint64_t w = 0, r = 0;
    if (n < 2) return n;                                   /* :217-220 */
    while (r < n) {
        if (r + 1 < n && t[r] == a && t[r + 1] == b) {     /* :231 */
            if (pc_get(tab, a, b, NULL)) pc_add(tab, a, b, -1, 0);      /* :240-246 */
            if (w > 0) {                                                 /* :248 */
                uint32_t p = t[w - 1];
                if (pc_get(tab, p, a, NULL)) pc_add(tab, p, a, -1, 0);  /* :249-256 */
                pc_add(tab, p, x, 1, 0);                                 /* :260 */
            }
            if (r + 2 < n) {                                             /* :263 */
                uint32_t y = t[r + 2];
                if (pc_get(tab, b, y, NULL)) pc_add(tab, b, y, -1, 0);  /* :264-270 */
                pc_add(tab, x, y, 1, 0);                                 /* :279 */
            }
            t[w++] = x;                                                  /* :236-237 */
            r += 2;
        } else {
            t[w++] = t[r++];                                             /* :291-296 */
        }
    }
    return w;
}

/* merge (no count maintenance), Tokenizer.h:162-199 */
static uint64_t merge_plain(uint32_t *t, uint64_t n, uint32_t a, uint32_t b, uint32_t x) {
    uint64_t w = 0, r = 0;
    while (r < n) {
        if (r + 1 < n && t[r] == a && t[r + 1] == b) { t[w++] = x; r += 2; }
        else t[w++] = t[r++];
    }
    return w;
}

/* merge_chunks, Tokenizer.h:309-320, plus the FIRST-mode recount :581-585 */
ORC_EXPORT void orc_merge(orc_state *st, uint32_t a, uint32_t b, uint32_t x) {
    for (uint64_t c = 0; c < st->n_chunks; c++) {
        uint32_t *t = st->tok + st->off[c];
        if (st->mode == 0) st->len[c] = merge_incremental(t, st->len[c], a, b, x, &st->tab);
        else               st->len[c] = merge_plain(t, st->len[c], a, b, x);
    }
    if (st->mode != 0) calculate_freqs(st);
}

/* The training loop, Tokenizer.h:557-589.  merges_out holds 2 u32 per merge,
 * counts_out (optional) the count printed by the verbose line :566-576.
 * Returns the number of merges performed (loop breaks only when the table
 * is empty, :586-588). */
ORC_EXPORT uint32_t orc_train_loop(orc_state *st, uint32_t vocab_size,
                                   uint32_t *merges_out, int32_t *counts_out) {
    uint32_t k = 0;
    for (uint32_t i = 256; i < vocab_size; i++) {
        uint32_t a, b; int32_t cnt;
        if (!orc_top(st, &a, &b, &cnt)) break;
        merges_out[2 * k] = a;
        merges_out[2 * k + 1] = b;
        if (counts_out) counts_out[k] = cnt;
        k++;
        orc_merge(st, a, b, i);
    }
    return k;
}

/* One-shot: Tokenizer::train, Tokenizer.h:489-598 (regex pre-split done by
 * the caller and passed as chunk_off). */
ORC_EXPORT uint32_t orc_train(const uint8_t *text, uint64_t n_bytes,
                              const uint64_t *chunk_off, uint64_t n_chunks,
                              uint32_t vocab_size, int mode,
                              uint32_t *merges_out, int32_t *counts_out) {
    orc_state *st = orc_create(text, n_bytes, chunk_off, n_chunks, mode);
    uint32_t k = orc_train_loop(st, vocab_size, merges_out, counts_out);
    orc_destroy(st);
    return k;
}

/* ------------------------------------------------------------------ */
/* introspection for step-level parity tests                           */
/* ------------------------------------------------------------------ */

ORC_EXPORT uint64_t orc_stream_len(orc_state *st) {
    uint64_t n = 0;
    for (uint64_t c = 0; c < st->n_chunks; c++) n += st->len[c];
    return n;
}

/* live tokens of all chunks back to back; chunk_len_out (optional) gets the
 * live length of every chunk */
ORC_EXPORT void orc_stream(orc_state *st, uint32_t *out, uint64_t *chunk_len_out) {
    uint64_t w = 0;
    for (uint64_t c = 0; c < st->n_chunks; c++) {
        memcpy(out + w, st->tok + st->off[c], st->len[c] * sizeof(uint32_t));
        w += st->len[c];
        if (chunk_len_out) chunk_len_out[c] = st->len[c];
    }
}

/* get_count(), PairCount.h:235: pairs ever inserted */
ORC_EXPORT uint64_t orc_table_size(orc_state *st) { return st->tab.n; }

/* get_all(), PairCount.h:271-278 (order unspecified) */
ORC_EXPORT uint64_t orc_table_dump(orc_state *st, uint32_t *a, uint32_t *b, int32_t *count) {
    uint64_t w = 0;
    for (uint64_t j = 0; j < st->tab.cap; j++) {
        if (!st->tab.slots[j].used) continue;
        a[w] = (uint32_t)(st->tab.slots[j].key >> 32);
        b[w] = (uint32_t)st->tab.slots[j].key;
        count[w] = st->tab.slots[j].count;
        w++;
    }
    return w;
}

/* get_pair(), PairCount.h:239-247; returns 0 when absent */
ORC_EXPORT int orc_get_pair(orc_state *st, uint32_t a, uint32_t b, int32_t *count) {
    return pc_get(&st->tab, a, b, count);
}

/* The byte-level initial histogram of calculate_freqs (Tokenizer.h:127-146)
 * as a dense 65,536-entry table, index = first*256 + second.  Chunks whose
 * NUL-quirk collapses them to one token contribute nothing. */
ORC_EXPORT void orc_pair_count_u8(const uint8_t *text, uint64_t n_bytes,
                                  const uint64_t *chunk_off, uint64_t n_chunks,
                                  uint32_t *table65536) {
    uint64_t one[2] = {0, n_bytes};
    if (!chunk_off) { chunk_off = one; n_chunks = 1; }
    memset(table65536, 0, 65536 * sizeof(uint32_t));
    for (uint64_t c = 0; c < n_chunks; c++) {
        const uint8_t *s = text + chunk_off[c];
        uint64_t n = chunk_off[c + 1] - chunk_off[c];
        long long id;
        if (n > 0 && s[0] == 0 && stoi_prefix(s + 1, n - 1, &id)) continue;
        for (uint64_t i = 0; i + 1 < n; i++)
            table65536[((uint32_t)s[i] << 8) | s[i + 1]]++;
    }
}

/* ------------------------------------------------------------------ */
/* PairCount interface KATs (test.cpp:15-106) drive the table directly */
/* ------------------------------------------------------------------ */

ORC_EXPORT pc_table *orc_pc_new(int lexical) {
    pc_table *t = (pc_table *)malloc(sizeof(pc_table));
    pc_init(t, lexical);
    return t;
}
ORC_EXPORT void orc_pc_free(pc_table *t) { pc_free(t); free(t); }
ORC_EXPORT int orc_pc_add(pc_table *t, uint32_t a, uint32_t b, int32_t f) { return pc_add(t, a, b, f, 0); }
ORC_EXPORT int orc_pc_get(pc_table *t, uint32_t a, uint32_t b, int32_t *c) { return pc_get(t, a, b, c); }
ORC_EXPORT uint64_t orc_pc_count(pc_table *t) { return t->n; }
ORC_EXPORT int orc_pc_top(pc_table *t, uint32_t *a, uint32_t *b, int32_t *c) {
    return t->use_heap ? pc_top_lexical(t, a, b, c) : pc_top_first(t, a, b, c);
}

/* ------------------------------------------------------------------ */
/* encode: internal_internal_encode, Tokenizer.h:325-367               */
/* One left-to-right pass replacing ANY pair found in merges_lookup    */
/* (first match wins, not rank order), repeated until a pass makes no  */
/* merge.  merges[2k],merges[2k+1] -> token 256+k (load(): :831-837;   */
/* a pair listed twice keeps the LAST index, as operator[] does).      */
/* ------------------------------------------------------------------ */

typedef struct { pc_table lut; } orc_encoder;

ORC_EXPORT orc_encoder *orc_encoder_new(const uint32_t *merges, uint32_t n_merges) {
    orc_encoder *e = (orc_encoder *)calloc(1, sizeof(orc_encoder));
    pc_init(&e->lut, 0);
    for (uint32_t k = 0; k < n_merges; k++) {
        uint32_t a = merges[2 * k], b = merges[2 * k + 1];
        pc_slot *s = pc_find(&e->lut, ((uint64_t)a << 32) | b);
        if (s) s->count = (int32_t)(256 + k);
        else pc_add(&e->lut, a, b, (int32_t)(256 + k), 1);
    }
    return e;
}
ORC_EXPORT void orc_encoder_free(orc_encoder *e) { pc_free(&e->lut); free(e); }

/* encodes one chunk in place; returns the new length */
ORC_EXPORT uint64_t orc_encode_chunk(orc_encoder *e, uint32_t *t, uint64_t n) {
    for (;;) {
        if (n < 2) return n;                                  /* :326-328 */
        uint64_t w = 0, r = 0, merged = 0;
        while (r < n) {
            int32_t id;
            if (r + 1 < n && pc_get(&e->lut, t[r], t[r + 1], &id)) {   /* :338-349 */
                t[w++] = (uint32_t)id; r += 2; merged++;
            } else {
                t[w++] = t[r++];                                         /* :352-358 */
            }
        }
        n = w;
        if (!merged) return n;                                /* :362-366 */
    }
}

/* text_to_vector exposed for the encode path (Tokenizer.h:669,701,708) */
ORC_EXPORT uint64_t orc_text_to_vector(const uint8_t *s, uint64_t n, uint32_t *out) {
    return text_to_vector(s, n, out);
}
