// Host-side mirror of MinBpeCC::Tokenizer::Tokenizer (reference
// code/include/Tokenizer.h:52-927): same method names, argument meaning and
// error behaviour, with train() running the lexical hot path on the GPU
// through the C-ABI of include/mbpe.h.
#ifndef MBPE_HOST_TOKENIZER_H
#define MBPE_HOST_TOKENIZER_H

#include <cstdint>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "mbpe_host.h"

namespace mbpe_host {

using Token = uint32_t;                       // Tokenizer.h:38
using TokenPair = std::pair<Token, Token>;    // Tokenizer.h:40

class Tokenizer {
public:
    enum CONFLICT_RESOLUTION { FIRST, LEXICAL };   // Tokenizer.h:54-57

    Tokenizer();
    explicit Tokenizer(const std::string &pattern);   // throws std::runtime_error like :427-431
    Tokenizer(const Tokenizer &) = delete;
    Tokenizer &operator=(const Tokenizer &) = delete;

    void set_special_tokens_from_file(const std::string &input_string);   // :476-486
    // :489-598; both CONFLICT_RESOLUTION values run on HIP device `device` (mbpe_train)
    void train(const std::string &text, int vocab_size, CONFLICT_RESOLUTION conflict_resolution, bool verbose,
               int device = 0);
    // :653-722; device >= 0 runs internal_encode (:325-377) on that HIP device instead of the host
    std::vector<Token> encode(const std::string &text, bool verbose, int device = -1);
    std::string decode(const std::vector<Token> &tokens, bool verbose);    // :725-751
    bool load(const std::string &path, bool verbose);                      // :754-872
    bool save(const std::string &path, bool write_vocab);                  // :875-926

    const std::vector<TokenPair> &get_merges() const { return merges_; }
    void set_merges(const std::vector<TokenPair> &m);
    const std::string &pattern() const { return pattern_; }

private:
    struct PairHash {
        size_t operator()(const TokenPair &k) const {          // pair_token_hash, :43-50
            size_t seed = 0;
            seed ^= std::hash<Token>()(k.first) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
            seed ^= std::hash<Token>()(k.second) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
            return seed;
        }
    };

    std::vector<Token> text_to_vector(const char *s, size_t n) const;       // :85-100
    void initialize_vocab();                                                // :103-111
    std::vector<std::string> split_on_special(const std::string &text) const;   // :605-650
    std::vector<Token> internal_internal_encode(std::vector<Token> text) const; // :325-367
    void rebuild_vocab();

    std::string pattern_;
    Splitter splitter_;
    // the reference keeps these in unordered_maps (:64-65); insertion order is kept here so
    // that save() is deterministic
    std::vector<std::pair<std::string, Token>> special_tokens_;
    std::unordered_map<Token, std::string> special_tokens_reverse_lookup_;
    std::unordered_map<TokenPair, Token, PairHash> merges_lookup_;
    std::vector<TokenPair> merges_;
    std::vector<std::vector<Token>> vocab_;
};

}  // namespace mbpe_host

#endif
