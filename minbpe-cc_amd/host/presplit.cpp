// Regex pre-split on the host: the part of Tokenizer::train that feeds the hot
// path (reference code/include/Tokenizer.h:500-540, compile options :407-415,
// JIT :435, match flags :512).  PCRE2 is bound at run time with dlopen because
// the image ships libpcre2-8.so.0 without its development header; the few
// entry points used are declared by hand below.
#include "mbpe.h"
#include "mbpe_host.h"

#include <dlfcn.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// --- the slice of the PCRE2 8-bit API this file uses ----------------------
constexpr uint32_t kPCRE2_CASELESS     = 0x00000008u;
constexpr uint32_t kPCRE2_UCP          = 0x00020000u;
constexpr uint32_t kPCRE2_UTF          = 0x00080000u;
constexpr uint32_t kPCRE2_NO_UTF_CHECK = 0x40000000u;
constexpr uint32_t kPCRE2_JIT_COMPLETE = 0x00000001u;
constexpr int      kPCRE2_ERROR_NOMATCH = -1;
constexpr uint32_t kPCRE2_INFO_MAXLOOKBEHIND = 15;

struct Pcre2Api {
    void *lib = nullptr;
    void *(*compile)(const uint8_t *, size_t, uint32_t, int *, size_t *, void *) = nullptr;
    void (*code_free)(void *) = nullptr;
    int (*jit_compile)(void *, uint32_t) = nullptr;
    void *(*match_data_create_from_pattern)(const void *, void *) = nullptr;
    void (*match_data_free)(void *) = nullptr;
    int (*match)(const void *, const uint8_t *, size_t, size_t, uint32_t, void *, void *) = nullptr;
    size_t *(*get_ovector_pointer)(void *) = nullptr;
    int (*get_error_message)(int, uint8_t *, size_t) = nullptr;
    int (*pattern_info)(const void *, uint32_t, void *) = nullptr;
    bool ok = false;
    std::string why;
};

Pcre2Api &pcre2() {
    static Pcre2Api api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"libpcre2-8.so.0", "libpcre2-8.so"};
        for (const char *n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) { api.why = "libpcre2-8.so.0 not found (needed for gpt2/gpt4 encoders)"; return; }
        auto sym = [&](const char *s) { return dlsym(api.lib, s); };
#define MBPE_BIND(field, name) \
        api.field = reinterpret_cast<decltype(api.field)>(sym(name)); \
        if (!api.field) { api.why = std::string("missing PCRE2 symbol ") + name; return; }
        MBPE_BIND(compile, "pcre2_compile_8")
        MBPE_BIND(code_free, "pcre2_code_free_8")
        MBPE_BIND(jit_compile, "pcre2_jit_compile_8")
        MBPE_BIND(match_data_create_from_pattern, "pcre2_match_data_create_from_pattern_8")
        MBPE_BIND(match_data_free, "pcre2_match_data_free_8")
        MBPE_BIND(match, "pcre2_match_8")
        MBPE_BIND(get_ovector_pointer, "pcre2_get_ovector_pointer_8")
        MBPE_BIND(get_error_message, "pcre2_get_error_message_8")
        MBPE_BIND(pattern_info, "pcre2_pattern_info_8")
#undef MBPE_BIND
        api.ok = true;
    });
    return api;
}

std::string pcre2_message(int code) {
    uint8_t buf[256] = {0};
    pcre2().get_error_message(code, buf, sizeof(buf));
    return std::string(reinterpret_cast<char *>(buf));
}

}  // namespace

namespace mbpe_host {

Splitter::~Splitter() { reset(); }

void Splitter::reset() {
    if (match_data_) pcre2().match_data_free(match_data_);
    if (code_) pcre2().code_free(code_);
    match_data_ = nullptr;
    code_ = nullptr;
}

// Tokenizer(const string &pattern), Tokenizer.h:391-451.
int Splitter::compile(const std::string &pattern, std::string *err) {
    reset();
    pattern_ = pattern;
    if (pattern.empty()) return MBPE_OK;   // "basic": no split, Tokenizer.h:400
    Pcre2Api &p = pcre2();
    if (!p.ok) { *err = p.why; return MBPE_ERR_REGEX; }
    uint32_t options = kPCRE2_UTF | kPCRE2_UCP;                          // :407
    if (pattern.find("(?i:") != std::string::npos) options |= kPCRE2_CASELESS;  // :413-415
    int errorcode = 0;
    size_t erroroffset = 0;
    code_ = p.compile(reinterpret_cast<const uint8_t *>(pattern.data()), pattern.size(), options,
                      &errorcode, &erroroffset, nullptr);
    if (!code_) {
        *err = "PCRE2 pattern compilation failed: " + pcre2_message(errorcode);   // :427-431
        return MBPE_ERR_REGEX;
    }
    p.jit_compile(code_, kPCRE2_JIT_COMPLETE);   // failure is not fatal, :435-442
    match_data_ = p.match_data_create_from_pattern(code_, nullptr);
    if (!match_data_) { *err = "PCRE2 match data creation failed."; return MBPE_ERR_REGEX; }
    return MBPE_OK;
}

// The match loop of Tokenizer::train / encode, Tokenizer.h:506-540 and :676-703:
// every non-empty match [start,end) is a chunk; an empty match advances the
// offset by one byte; NOMATCH ends the loop.  Runs from `offset` until the offset
// reaches `stop` (n for the whole text); `offset` is left where the loop stood.
// Returns MBPE_OK, or 1 when NOMATCH ended the loop (no further match to the end of the text).
static int match_loop(const void *code, void *match_data, const uint8_t *text, uint64_t n, size_t &offset,
                      uint64_t stop, std::vector<uint64_t> *starts, std::vector<uint64_t> *ends, std::string *err) {
    Pcre2Api &p = pcre2();
    while (offset < stop || stop == n) {
        int rc = p.match(code, text, n, offset, kPCRE2_NO_UTF_CHECK, match_data, nullptr);
        if (rc < 0) {
            if (rc == kPCRE2_ERROR_NOMATCH) return 1;
            *err = "PCRE2 match error: " + pcre2_message(rc);            // :517-522
            return MBPE_ERR_REGEX;
        }
        size_t *ov = p.get_ovector_pointer(match_data);
        size_t start = ov[0], end = ov[1];
        if (start == end) {                                              // :529-533
            if (offset >= n) return 1;
            offset++;
            continue;
        }
        starts->push_back(start);
        ends->push_back(end);
        offset = end;
    }
    return MBPE_OK;
}

int Splitter::split(const uint8_t *text, uint64_t n, std::vector<uint64_t> *starts,
                    std::vector<uint64_t> *ends, std::string *err) const {
    starts->clear();
    ends->clear();
    if (!code_) {                 // no pattern: one chunk, :541-544
        starts->push_back(0);
        ends->push_back(n);
        return MBPE_OK;
    }
    Pcre2Api &p = pcre2();
    static const uint8_t kEmpty[1] = {0};
    if (!text) text = kEmpty;      // std::string::data() of an empty string is never NULL

    // Large texts are split by several host threads.  A match attempt at offset o depends only on
    // the text from o on when the pattern looks behind nowhere (PCRE2_INFO_MAXLOOKBEHIND == 0:
    // true of the gpt2 / gpt4 patterns), so the sequential loop of the reference is a walk over
    // offsets whose continuation is determined by the current offset alone.  Thread k runs the loop
    // from a guessed offset S_k; the true walk, stitched sequentially, takes over thread k's matches
    // from the first offset both have visited (normally a handful of matches after S_k), and keeps
    // matching on its own until then.  The result is the sequential one, match for match.
    uint32_t lookbehind = 1;
    if (p.pattern_info(code_, kPCRE2_INFO_MAXLOOKBEHIND, &lookbehind) != 0) lookbehind = 1;
    unsigned n_threads = 1;
    if (lookbehind == 0 && n >= (8u << 20)) {
        unsigned hw = std::thread::hardware_concurrency();
        const char *env = getenv("MBPE_SPLIT_THREADS");
        unsigned want = env ? (unsigned)atoi(env) : 16u;
        n_threads = std::max(1u, std::min({want, hw ? hw : 1u, (unsigned)(n >> 20)}));
    }
    if (n_threads <= 1) {
        size_t offset = 0;
        int rc = match_loop(code_, match_data_, text, n, offset, n, starts, ends, err);
        return rc < 0 ? rc : MBPE_OK;
    }

    struct Part {
        uint64_t s0 = 0;                       // guessed start offset (a character boundary)
        std::vector<uint64_t> starts, ends;
        size_t final_offset = 0;
        bool no_more = false;                  // NOMATCH: nothing matches from final_offset to the end
        int rc = MBPE_OK;
        std::string err;
    };
    std::vector<Part> parts(n_threads);
    for (unsigned k = 0; k < n_threads; ++k) {
        uint64_t s0 = (uint64_t)k * n / n_threads;
        while (s0 < n && (text[s0] & 0xC0) == 0x80) ++s0;     // not inside a UTF-8 sequence
        parts[k].s0 = s0;
    }
    std::vector<std::thread> pool;
    for (unsigned k = 0; k < n_threads; ++k) {
        pool.emplace_back([&, k] {
            Part &pt = parts[k];
            void *md = k == 0 ? match_data_ : p.match_data_create_from_pattern(code_, nullptr);
            if (!md) { pt.rc = MBPE_ERR_REGEX; pt.err = "PCRE2 match data creation failed."; return; }
            const uint64_t stop = k + 1 < n_threads ? parts[k + 1].s0 : n;
            size_t offset = pt.s0;
            pt.starts.reserve((stop - pt.s0) / 3);
            pt.ends.reserve((stop - pt.s0) / 3);
            int rc = match_loop(code_, md, text, n, offset, stop, &pt.starts, &pt.ends, &pt.err);
            if (rc < 0) pt.rc = rc;
            pt.no_more = rc == 1;
            pt.final_offset = offset;
            if (k != 0) p.match_data_free(md);
        });
    }
    for (auto &t : pool) t.join();
    for (const Part &pt : parts)
        if (pt.rc != MBPE_OK) { *err = pt.err; return pt.rc; }

    // stitch: `cur` is the offset of the true walk.  The result is a list of segments (a range of a
    // thread's matches, or a few matches the true walk made on its own between two threads),
    // copied into place by the threads afterwards.
    struct Seg { const std::vector<uint64_t> *s, *e; size_t from, to; };
    std::vector<Seg> segs;
    std::vector<std::vector<uint64_t>> bridge_s(n_threads), bridge_e(n_threads);
    segs.push_back({&parts[0].starts, &parts[0].ends, 0, parts[0].starts.size()});
    size_t cur = parts[0].final_offset;
    bool done = parts[0].no_more;
    for (unsigned k = 1; k < n_threads && !done; ++k) {
        Part &pt = parts[k];
        const uint64_t region_end = k + 1 < n_threads ? parts[k + 1].s0 : n;
        for (;;) {
            // has thread k stood at offset cur?  (its start, or the end of one of its matches)
            size_t from = SIZE_MAX;
            if (cur == pt.s0) from = 0;
            else {
                auto it = std::lower_bound(pt.ends.begin(), pt.ends.end(), (uint64_t)cur);
                if (it != pt.ends.end() && *it == cur) from = (size_t)(it - pt.ends.begin()) + 1;
            }
            if (from != SIZE_MAX) {
                segs.push_back({&pt.starts, &pt.ends, from, pt.starts.size()});
                cur = pt.final_offset;
                done = pt.no_more;
                break;
            }
            if (cur >= region_end) break;      // walked through the whole region without meeting thread k
            // one more step of the true walk
            size_t off = cur;
            int rc = match_loop(code_, match_data_, text, n, off, std::min<uint64_t>(cur + 1, n), &bridge_s[k],
                                &bridge_e[k], err);
            if (rc < 0) return rc;
            if (rc == 1) { done = true; break; }
            cur = off;
        }
        if (!bridge_s[k].empty()) {
            // (the bridge precedes the part of thread k that was adopted, if any)
            Seg b{&bridge_s[k], &bridge_e[k], 0, bridge_s[k].size()};
            if (segs.back().s == &pt.starts) segs.insert(segs.end() - 1, b);
            else segs.push_back(b);
        }
    }
    std::vector<uint64_t> tail_s, tail_e;
    if (!done && cur < n) {
        // (only when the last region never met the true walk)
        size_t off = cur;
        int rc = match_loop(code_, match_data_, text, n, off, n, &tail_s, &tail_e, err);
        if (rc < 0) return rc;
        segs.push_back({&tail_s, &tail_e, 0, tail_s.size()});
    }
    size_t total = 0;
    std::vector<size_t> at(segs.size());
    for (size_t i = 0; i < segs.size(); ++i) { at[i] = total; total += segs[i].to - segs[i].from; }
    starts->reserve(total + 1);       // (+1: mbpe_presplit appends the end of the text)
    starts->resize(total);
    ends->resize(total);
    std::vector<std::thread> copiers;
    for (size_t i = 0; i < segs.size(); ++i)
        copiers.emplace_back([&, i] {
            const Seg &g = segs[i];
            std::copy(g.s->begin() + g.from, g.s->begin() + g.to, starts->begin() + at[i]);
            std::copy(g.e->begin() + g.from, g.e->begin() + g.to, ends->begin() + at[i]);
        });
    for (auto &t : copiers) t.join();
    return MBPE_OK;
}

const char *split_pattern_for(const std::string &encoder) {
    // Tokenizer.h:59-60
    static const char *gpt2 =
        "'(?:[sdmt]|ll|ve|re)| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+";
    static const char *gpt4 =
        "'(?i:[sdmt]|ll|ve|re)|[^\\r\\n\\p{L}\\p{N}]?+\\p{L}+|\\p{N}{1,3}| ?[^\\s\\p{L}\\p{N}]++[\\r\\n]*|\\s*[\\r\\n]|\\s+(?!\\S)|\\s+";
    if (encoder == "gpt2") return gpt2;
    if (encoder == "gpt4") return gpt4;
    if (encoder == "basic") return "";
    return nullptr;
}

}  // namespace mbpe_host

// ---- C-ABI ---------------------------------------------------------------

struct mbpe_split {
    std::vector<uint64_t> starts, ends;   // chunk c = [starts[c], ends[c])
    std::vector<uint64_t> off;            // n_chunks + 1 offsets when the chunks tile the text, else empty
    bool gaps = false;
};

extern "C" {

int mbpe_presplit(const char *pattern, const uint8_t *text, uint64_t n_bytes, mbpe_split **out) {
    if (!pattern || (!text && n_bytes) || !out) {
        mbpe_host::set_last_error("mbpe_presplit: NULL argument");
        return MBPE_ERR_ARG;
    }
    std::string err;
    mbpe_host::Splitter sp;
    int rc = sp.compile(pattern, &err);
    if (rc != MBPE_OK) { mbpe_host::set_last_error(err); return rc; }
    mbpe_split *s = new mbpe_split();
    rc = sp.split(text, n_bytes, &s->starts, &s->ends, &err);
    if (rc != MBPE_OK) { mbpe_host::set_last_error(err); delete s; return rc; }
    // The gpt2/gpt4 patterns match every byte of valid UTF-8 input; any other pattern may leave
    // bytes between matches, which the reference simply skips (Tokenizer.h:506-540).
    uint64_t pos = 0;
    for (size_t i = 0; i < s->starts.size(); i++) {
        if (s->starts[i] != pos) s->gaps = true;
        pos = s->ends[i];
    }
    if (pos != n_bytes) s->gaps = true;
    if (!s->gaps) {
        s->off = s->starts;           // chunk c starts where chunk c - 1 ended
        s->off.push_back(n_bytes);
        if (s->starts.empty()) { s->off.clear(); s->off.push_back(0); }
    }
    *out = s;
    return MBPE_OK;
}

uint64_t mbpe_split_count(const mbpe_split *s) { return s ? s->starts.size() : 0; }
const uint64_t *mbpe_split_offsets(const mbpe_split *s) {
    if (!s) return nullptr;
    if (s->gaps) {
        mbpe_host::set_last_error("the split pattern left bytes unmatched: use mbpe_split_starts / mbpe_split_ends "
                                  "with mbpe_load_corpus_ranges");
        return nullptr;
    }
    return s->off.data();
}
int mbpe_split_has_gaps(const mbpe_split *s) { return s && s->gaps ? 1 : 0; }
const uint64_t *mbpe_split_starts(const mbpe_split *s) { return s ? s->starts.data() : nullptr; }
const uint64_t *mbpe_split_ends(const mbpe_split *s) { return s ? s->ends.data() : nullptr; }
void mbpe_split_free(mbpe_split *s) { delete s; }

const char *mbpe_split_pattern(const char *encoder_name) {
    if (!encoder_name) return nullptr;
    return mbpe_host::split_pattern_for(encoder_name);
}

}  // extern "C"
