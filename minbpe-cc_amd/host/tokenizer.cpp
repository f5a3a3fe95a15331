// Host-side mirror of the reference Tokenizer (see tokenizer.h).  Everything
// here is host bookkeeping around the hot path: model files, special tokens,
// encode / decode.  train() hands the byte buffer and the chunk offsets to the
// GPU through mbpe_train_lexical.
#include "tokenizer.h"

#include "mbpe.h"
#include "mbpe_tokenizer.h"

#include <cstdio>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>

namespace mbpe_host {

Tokenizer::Tokenizer() {}

Tokenizer::Tokenizer(const std::string &pattern) : pattern_(pattern) {
    std::string err;
    if (splitter_.compile(pattern, &err) != MBPE_OK) throw std::runtime_error(err);   // Tokenizer.h:427-431
}

// Tokenizer.h:85-100, with the NUL quirk :86-93 (std::stoi on the remainder).
std::vector<Token> Tokenizer::text_to_vector(const char *s, size_t n) const {
    if (n > 0 && s[0] == '\0') {
        try {
            int id = std::stoi(std::string(s + 1, n - 1));
            return std::vector<Token>{static_cast<Token>(id)};
        } catch (...) {
        }
    }
    std::vector<Token> out;
    out.reserve(n);
    for (size_t i = 0; i < n; ++i) out.push_back(static_cast<uint8_t>(s[i]));   // char_to_token :80-82
    return out;
}

void Tokenizer::initialize_vocab() {   // :103-111
    vocab_.clear();
    vocab_.reserve(256);
    for (int i = 0; i < 256; i++) vocab_.push_back(std::vector<Token>{static_cast<Token>(i)});
}

void Tokenizer::rebuild_vocab() {      // :843-861 / :562-564
    initialize_vocab();
    for (const auto &m : merges_) {
        // ids beyond the vocabulary built so far (hand-edited model) expand to nothing
        std::vector<Token> appended;
        if (m.first < vocab_.size()) appended = vocab_[m.first];
        if (m.second < vocab_.size()) appended.insert(appended.end(), vocab_[m.second].begin(), vocab_[m.second].end());
        vocab_.push_back(appended);
    }
}

void Tokenizer::set_merges(const std::vector<TokenPair> &m) {
    merges_ = m;
    merges_lookup_.clear();
    Token idx = 256;
    for (const auto &p : merges_) merges_lookup_[p] = idx++;   // operator[]: a repeated pair keeps the last id
    rebuild_vocab();
}

void Tokenizer::set_special_tokens_from_file(const std::string &input_string) {   // :476-486
    special_tokens_.clear();
    special_tokens_reverse_lookup_.clear();
    std::istringstream iss(input_string);
    std::string key;
    Token value;
    while (iss >> key >> value) {
        bool found = false;
        for (auto &kv : special_tokens_)
            if (kv.first == key) { kv.second = value; found = true; }
        if (!found) special_tokens_.emplace_back(key, value);
        special_tokens_reverse_lookup_[value] = key;
    }
}

void Tokenizer::train(const std::string &text, int vocab_size, CONFLICT_RESOLUTION conflict_resolution,
                      bool verbose, int device) {
    if (vocab_size < 256) throw std::runtime_error("vocab_size must be >= 256");   // assert, :492
    merges_.clear();
    merges_lookup_.clear();
    initialize_vocab();

    std::vector<uint64_t> starts, ends;
    std::string err;
    const uint8_t *bytes = reinterpret_cast<const uint8_t *>(text.data());
    if (splitter_.split(bytes, text.size(), &starts, &ends, &err) != MBPE_OK) throw std::runtime_error(err);
    if (verbose) std::cout << "Split input text into " << starts.size() << " chunks\n";   // :546-548
    const bool chunked = splitter_.has_pattern();

    mbpe_ctx *ctx = nullptr;
    if (mbpe_create(device, &ctx) != MBPE_OK) throw std::runtime_error(mbpe_last_error());
    const uint32_t cap = static_cast<uint32_t>(vocab_size - 256);
    std::vector<uint32_t> flat(2 * static_cast<size_t>(cap) + 2);
    std::vector<int32_t> had(cap + 1);
    uint32_t n_merges = 0;
    mbpe_stats st;
    // chunks as ranges: bytes between two matches belong to no chunk, as in the reference's loop (:506-540)
    int rc = chunked ? mbpe_load_corpus_ranges(ctx, bytes, text.size(), starts.data(), ends.data(), starts.size(), 0)
                     : mbpe_load_corpus(ctx, bytes, text.size(), nullptr, 0, 0);
    if (rc == MBPE_OK) rc = mbpe_set_option(ctx, "conflict_resolution", conflict_resolution == LEXICAL ? 1 : 0);
    if (rc == MBPE_OK) rc = mbpe_train_begin(ctx, static_cast<uint32_t>(vocab_size));
    if (rc == MBPE_OK) rc = mbpe_train_steps(ctx, cap, nullptr);
    if (rc == MBPE_OK) rc = mbpe_train_result(ctx, flat.data(), had.data(), cap, &n_merges);
    if (rc == MBPE_OK) rc = mbpe_get_stats(ctx, &st);
    std::string msg = rc == MBPE_OK ? "" : mbpe_last_error();
    mbpe_destroy(ctx);
    if (rc != MBPE_OK) throw std::runtime_error(msg);

    const int total_merges = vocab_size - 256;
    for (uint32_t k = 0; k < n_merges; ++k) {            // the bookkeeping of :562-579
        const TokenPair mp{flat[2 * k], flat[2 * k + 1]};
        std::vector<Token> appended{vocab_[mp.first]};
        appended.insert(appended.end(), vocab_[mp.second].begin(), vocab_[mp.second].end());
        vocab_.push_back(appended);
        if (verbose) {                                   // :565-577
            std::string s;
            for (auto c : appended) s += (c >= 32 && c < 127) ? static_cast<char>(c) : ' ';
            std::cout << "merge " << k + 1 << "/" << total_merges << ": (" << mp.first << ", " << mp.second
                      << ") -> " << 256 + k << " (b'" << s << "') had " << had[k] << " occurrences\n";
        }
        merges_.push_back(mp);
        merges_lookup_[mp] = 256 + k;
    }
    if (verbose)                                         // :591-597
        std::cout << "Length of training text " << text.length() << ". After merges " << st.n_live << ".\n";
}

// :605-650
// :605-650.  Same result as the reference's rescan-from-the-cursor loop, computed in one sweep: every occurrence of
// every special token is listed once, sorted by (position, token order), and an occurrence is taken when it starts
// at or after the end of the previous one taken.  A taken token becomes the marker "\0<id>" (:635-637).
std::vector<std::string> Tokenizer::split_on_special(const std::string &text) const {
    std::vector<std::string> parts;
    struct Occ { size_t pos; size_t tok; };
    std::vector<Occ> occ;
    for (size_t k = 0; k < special_tokens_.size(); ++k) {
        const std::string &name = special_tokens_[k].first;
        if (name.empty()) continue;
        for (size_t p = text.find(name); p != std::string::npos; p = text.find(name, p + 1)) occ.push_back({p, k});
    }
    std::sort(occ.begin(), occ.end(), [](const Occ &a, const Occ &b) { return a.pos != b.pos ? a.pos < b.pos : a.tok < b.tok; });
    size_t cursor = 0;
    for (const Occ &o : occ) {
        if (o.pos < cursor) continue;                      // inside a token already taken
        if (o.pos > cursor) parts.push_back(text.substr(cursor, o.pos - cursor));
        parts.push_back(std::string(1, '\0') + std::to_string(special_tokens_[o.tok].second));
        cursor = o.pos + special_tokens_[o.tok].first.size();
    }
    if (cursor < text.size()) parts.push_back(text.substr(cursor));
    if (parts.empty()) parts.push_back(text);
    return parts;
}

// :325-367: one left-to-right pass replacing ANY pair found in merges_lookup
// (first match wins, not rank order), repeated until a pass changes nothing.
std::vector<Token> Tokenizer::internal_internal_encode(std::vector<Token> text) const {
    for (;;) {
        if (text.size() < 2) return text;
        std::vector<Token> out;
        out.reserve(text.size());
        size_t i = 0, merge_count = 0;
        const size_t len = text.size();
        while (i < len) {
            bool merged = false;
            if (i + 1 < len) {
                auto it = merges_lookup_.find(TokenPair{text[i], text[i + 1]});
                if (it != merges_lookup_.end()) {
                    out.push_back(it->second);
                    i += 2;
                    merge_count++;
                    merged = true;
                }
            }
            if (!merged) out.push_back(text[i++]);
        }
        if (merge_count == 0) return out;
        text.swap(out);
    }
}

// :653-722.  device < 0: internal_encode on the host (below); device >= 0: on that HIP device (mbpe_encode_chunks)
std::vector<Token> Tokenizer::encode(const std::string &text, bool verbose, int device) {
    auto split_text = split_on_special(text);
    if (verbose) {
        std::cout << "Splitting input text into " << split_text.size() << " parts\n";
        for (const auto &part : split_text) {
            bool is_special = part.size() > 0 && part[0] == '\0';
            std::cout << "Part: \"" << part << "\" special: " << is_special << "\n";
        }
    }
    // the chunks, as one byte buffer + offsets: special markers and, with a pattern, the regex matches of every
    // other part (:664-704); without one, every part is a chunk (:706-709)
    std::string buf;
    std::vector<uint64_t> off{0};
    buf.reserve(text.size() + 16 * split_text.size());
    for (const auto &part : split_text) {
        if (splitter_.has_pattern() && !(part.size() > 0 && part[0] == '\0')) {
            std::vector<uint64_t> starts, ends;
            std::string err;
            if (splitter_.split(reinterpret_cast<const uint8_t *>(part.data()), part.size(), &starts, &ends, &err) !=
                MBPE_OK)
                throw std::runtime_error(err);                   // :686-691
            for (size_t i = 0; i < starts.size(); ++i) {
                buf.append(part, starts[i], ends[i] - starts[i]);
                off.push_back(buf.size());
            }
        } else {
            buf += part;
            off.push_back(buf.size());
        }
    }
    std::vector<Token> out;
    if (device >= 0) {
        std::vector<uint32_t> flat;
        flat.reserve(2 * merges_.size());
        for (const auto &m : merges_) { flat.push_back(m.first); flat.push_back(m.second); }
        out.resize(buf.size());
        uint64_t n = 0;
        const int rc = mbpe_encode_chunks(device, reinterpret_cast<const uint8_t *>(buf.data()), buf.size(), off.data(),
                                          off.size() - 1, flat.data(), static_cast<uint32_t>(merges_.size()),
                                          out.data(), out.size(), &n, nullptr);
        if (rc != MBPE_OK) throw mbpe_host::CodedError(rc, mbpe_last_error());       // (the C-ABI hands the code on unchanged)
        out.resize(n);
    } else {
        for (size_t c = 0; c + 1 < off.size(); ++c) {           // internal_encode :370-377 + flatten :713-717
            auto enc = internal_internal_encode(text_to_vector(buf.data() + off[c], off[c + 1] - off[c]));
            out.insert(out.end(), enc.begin(), enc.end());
        }
    }
    if (verbose) std::cout << "Encoded input text (length " << text.length() << ") to " << out.size() << " tokens\n";
    return out;
}

// :725-751
std::string Tokenizer::decode(const std::vector<Token> &tokens, bool verbose) {
    if (verbose) std::cout << "Decoding " << tokens.size() << " tokens\n";
    std::string text;
    for (Token tkn : tokens) {
        auto sp = special_tokens_reverse_lookup_.find(tkn);
        if (sp != special_tokens_reverse_lookup_.end()) { text += sp->second; continue; }
        if (tkn >= vocab_.size()) {
            std::cerr << "Warning: Attempted to decode invalid token ID: " << tkn << "\n";
            continue;
        }
        for (Token c : vocab_[tkn]) text.push_back(static_cast<char>(c));
    }
    return text;
}

// :754-872
bool Tokenizer::load(const std::string &path, bool verbose) {
    std::ifstream in(path, std::ios::in);
    if (!in.is_open()) {
        std::cerr << "Failed to open file for loading: " << path << "\n";
        return false;
    }
    std::string version;
    std::getline(in, version);
    if (version != "minbpe v1") {
        std::cerr << "Unexpected version: " << version << "\n";
        return false;
    }
    merges_lookup_.clear();
    merges_.clear();
    initialize_vocab();
    std::getline(in, pattern_);
    {
        std::string err;   // recompile the pattern of the model file (:770-814)
        if (splitter_.compile(pattern_, &err) != MBPE_OK) {
            std::cerr << "PCRE2 compilation failed on load: " << err << "\n";
            return false;
        }
    }
    int num_special = 0;
    in >> num_special;
    for (int i = 0; i < num_special; i++) {                     // :817-828 (adds to the existing specials)
        std::string token;
        Token id;
        in >> token >> id;
        bool found = false;
        for (auto &kv : special_tokens_)
            if (kv.first == token) { kv.second = id; found = true; }
        if (!found) special_tokens_.emplace_back(token, id);
        special_tokens_reverse_lookup_[id] = token;
        if (verbose) std::cout << "Loaded special token: " << token << " with ID " << id << "\n";
    }
    Token idx1, idx2, cur = 256;
    while (in >> idx1 >> idx2) {                                // :831-837
        merges_.push_back({idx1, idx2});
        merges_lookup_[{idx1, idx2}] = cur++;
    }
    if (verbose) std::cout << "Read input model from \"" << path << "\"\n";
    rebuild_vocab();
    if (verbose)
        std::cout << "Loaded vocab with " << merges_.size() << " merges, vocab size is " << vocab_.size() << "\n";
    return true;
}

// :875-926
bool Tokenizer::save(const std::string &path, bool write_vocab) {
    std::ofstream out(path, std::ios::out);
    if (!out.is_open()) {
        std::cerr << "Unable to open file for saving: " << path << std::endl;
        return false;
    }
    std::cout << "Writing model...\n";
    out << "minbpe v1" << std::endl;
    out << pattern_ << std::endl;
    out << special_tokens_.size() << std::endl;
    for (const auto &st : special_tokens_) out << st.first << ' ' << st.second << std::endl;
    for (const auto &m : merges_) out << m.first << ' ' << m.second << "\n";
    out.close();
    if (write_vocab) {                                          // :894-918
        std::ofstream vf(path + ".vocab", std::ios::out);
        if (!vf.is_open()) {
            std::cerr << "Failed to open .vocab file for writing: " << path + ".vocab" << std::endl;
            return false;
        }
        Token id = 0;
        for (const auto &v : vocab_) {
            vf << std::setw(6) << std::left << id << ": \"";
            for (Token c : v) {
                if (c >= 32 && c <= 126) vf << static_cast<char>(c);
                else vf << "\xEF\xBF\xBD";                      // U+FFFD
            }
            vf << "\"\n";
            id++;
        }
    }
    std::cout << "Complete.\n";
    return true;
}

}  // namespace mbpe_host

// ---- C-ABI (include/mbpe_tokenizer.h) ----------------------------------------

struct mbpe_tokenizer {
    mbpe_host::Tokenizer *t;
};

namespace {
template <typename F>
int guarded(F f) {
    try {
        return f();
    } catch (const std::exception &e) {
        mbpe_host::set_last_error(e.what());
        return MBPE_ERR_ARG;
    }
}
}  // namespace

extern "C" {

int mbpe_tok_create(const char *pattern, mbpe_tokenizer **out) {
    if (!pattern || !out) { mbpe_host::set_last_error("mbpe_tok_create: NULL argument"); return MBPE_ERR_ARG; }
    try {
        *out = new mbpe_tokenizer{new mbpe_host::Tokenizer(pattern)};
        return MBPE_OK;
    } catch (const std::exception &e) {
        mbpe_host::set_last_error(e.what());
        return MBPE_ERR_REGEX;
    }
}

void mbpe_tok_destroy(mbpe_tokenizer *t) {
    if (!t) return;
    delete t->t;
    delete t;
}

int mbpe_tok_set_special_tokens(mbpe_tokenizer *t, const char *text, uint64_t n) {
    if (!t || (!text && n)) return MBPE_ERR_ARG;
    return guarded([&] { t->t->set_special_tokens_from_file(std::string(text ? text : "", n)); return (int)MBPE_OK; });
}

int mbpe_tok_train(mbpe_tokenizer *t, const uint8_t *text, uint64_t n, uint32_t vocab_size, int conflict_resolution,
                   int verbose, int device_id) {
    if (!t || (!text && n)) return MBPE_ERR_ARG;
    return guarded([&] {
        t->t->train(std::string(reinterpret_cast<const char *>(text), n), (int)vocab_size,
                    conflict_resolution ? mbpe_host::Tokenizer::LEXICAL : mbpe_host::Tokenizer::FIRST, verbose != 0,
                    device_id);
        return (int)MBPE_OK;
    });
}

int mbpe_tok_set_merges(mbpe_tokenizer *t, const uint32_t *merges, uint32_t n_merges) {
    if (!t || (!merges && n_merges)) return MBPE_ERR_ARG;
    std::vector<mbpe_host::TokenPair> m;
    for (uint32_t k = 0; k < n_merges; ++k) m.push_back({merges[2 * k], merges[2 * k + 1]});
    t->t->set_merges(m);
    return MBPE_OK;
}

int mbpe_tok_get_merges(mbpe_tokenizer *t, uint32_t *merges_out, uint32_t cap, uint32_t *n_out) {
    if (!t || !n_out) return MBPE_ERR_ARG;
    const auto &m = t->t->get_merges();
    *n_out = (uint32_t)m.size();
    if (!merges_out) return MBPE_OK;
    if (cap < m.size()) { mbpe_host::set_last_error("merges_out too small"); return MBPE_ERR_ARG; }
    for (size_t k = 0; k < m.size(); ++k) { merges_out[2 * k] = m[k].first; merges_out[2 * k + 1] = m[k].second; }
    return MBPE_OK;
}

int mbpe_tok_save(mbpe_tokenizer *t, const char *path, int write_vocab) {
    if (!t || !path) return MBPE_ERR_ARG;
    return guarded([&] { return t->t->save(path, write_vocab != 0) ? (int)MBPE_OK : (int)MBPE_ERR_IO; });
}

int mbpe_tok_load(mbpe_tokenizer *t, const char *path, int verbose) {
    if (!t || !path) return MBPE_ERR_ARG;
    return guarded([&] { return t->t->load(path, verbose != 0) ? (int)MBPE_OK : (int)MBPE_ERR_IO; });
}

int mbpe_tok_encode(mbpe_tokenizer *t, const uint8_t *text, uint64_t n, int verbose, uint32_t *tokens_out,
                    uint64_t cap, uint64_t *n_out) {
    if (!t || (!text && n) || !n_out) return MBPE_ERR_ARG;
    return guarded([&] {
        auto enc = t->t->encode(std::string(reinterpret_cast<const char *>(text), n), verbose != 0);
        *n_out = enc.size();
        if (!tokens_out) return (int)MBPE_OK;
        if (cap < enc.size()) { mbpe_host::set_last_error("tokens_out too small"); return (int)MBPE_ERR_ARG; }
        memcpy(tokens_out, enc.data(), enc.size() * sizeof(uint32_t));
        return (int)MBPE_OK;
    });
}

int mbpe_tok_encode_device(mbpe_tokenizer *t, const uint8_t *text, uint64_t n, int verbose, int device_id,
                           uint32_t *tokens_out, uint64_t cap, uint64_t *n_out) {
    if (!t || (!text && n) || !n_out || device_id < 0) return MBPE_ERR_ARG;
    try {
        auto enc = t->t->encode(std::string(reinterpret_cast<const char *>(text), n), verbose != 0, device_id);
        *n_out = enc.size();
        if (!tokens_out) return MBPE_OK;
        if (cap < enc.size()) { mbpe_host::set_last_error("tokens_out too small"); return MBPE_ERR_ARG; }
        memcpy(tokens_out, enc.data(), enc.size() * sizeof(uint32_t));
        return MBPE_OK;
    } catch (const mbpe_host::CodedError &e) {      // mbpe_encode_chunks failed: its own code (no device, memory, HIP, ids)
        mbpe_host::set_last_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        mbpe_host::set_last_error(e.what());
        return MBPE_ERR_ARG;
    }
}

int mbpe_tok_decode(mbpe_tokenizer *t, const uint32_t *tokens, uint64_t n, int verbose, uint8_t *bytes_out,
                    uint64_t cap, uint64_t *n_out) {
    if (!t || (!tokens && n) || !n_out) return MBPE_ERR_ARG;
    return guarded([&] {
        auto s = t->t->decode(std::vector<mbpe_host::Token>(tokens, tokens + n), verbose != 0);
        *n_out = s.size();
        if (!bytes_out) return (int)MBPE_OK;
        if (cap < s.size()) { mbpe_host::set_last_error("bytes_out too small"); return (int)MBPE_ERR_ARG; }
        memcpy(bytes_out, s.data(), s.size());
        return (int)MBPE_OK;
    });
}

}  // extern "C"
