// Host-side C++ layer above the C-ABI (include/mbpe.h): the pieces of the
// reference's Tokenizer that surround the hot path -- regex pre-split, the
// "minbpe v1" model file, encode / decode -- mirrored with the reference's
// names and argument meaning so that callers and tests read like the
// reference's own (code/include/Tokenizer.h:379-927).
#ifndef MBPE_HOST_H
#define MBPE_HOST_H

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace mbpe_host {

// last error text of the calling thread (returned by mbpe_last_error())
void set_last_error(const std::string &msg);
const char *last_error();

// a failed C-ABI call inside the C++ layer: the MBPE_ERR_* code travels with the text
struct CodedError : std::runtime_error {
    int code;
    CodedError(int c, const std::string &msg) : std::runtime_error(msg), code(c) {}
};

// Tokenizer.h:59-60; nullptr for an unknown encoder name
const char *split_pattern_for(const std::string &encoder);

// Compiled split pattern + match loop (Tokenizer.h:391-451, :506-540).
class Splitter {
public:
    Splitter() = default;
    ~Splitter();
    Splitter(const Splitter &) = delete;
    Splitter &operator=(const Splitter &) = delete;

    int compile(const std::string &pattern, std::string *err);   // replaces any earlier pattern
    void reset();
    // chunk c = [starts[c], ends[c]); empty pattern -> one chunk = whole text
    int split(const uint8_t *text, uint64_t n, std::vector<uint64_t> *starts,
              std::vector<uint64_t> *ends, std::string *err) const;
    bool has_pattern() const { return code_ != nullptr; }
    const std::string &pattern() const { return pattern_; }

private:
    std::string pattern_;
    void *code_ = nullptr;
    void *match_data_ = nullptr;
};

}  // namespace mbpe_host

#endif
