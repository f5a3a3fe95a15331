// minbpe-cc: the reference's command line (code/examples/minbpe-cc.cpp:92-265)
// on top of the MI355X training path.  Same flags, defaults, messages and exit
// codes; `--train -c lexical` runs on the GPU.  The only addition is --device.
#include "mbpe.h"
#include "mbpe_tokenizer.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <vector>

using Token = uint32_t;   // Tokenizer.h:38

namespace {

bool exists(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0;
}

bool load_file_to_string(const std::string &path, std::string *out, std::string *err) {   // :36-46
    std::ifstream file(path);
    if (!file) { *err = std::strerror(errno); return false; }
    std::stringstream buffer;
    buffer << file.rdbuf();
    *out = buffer.str();
    return true;
}

bool save_encoding(const std::string &path, const std::vector<Token> &encoded, std::string *err) {   // :58-69
    std::ofstream file(path, std::ios::binary);
    if (!file) { *err = std::strerror(errno); return false; }
    for (const auto &code : encoded) file.write(reinterpret_cast<const char *>(&code), sizeof(Token));
    return true;
}

bool load_encoding(const std::string &path, std::vector<Token> *data, std::string *err) {   // :71-89
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) { *err = std::strerror(errno); return false; }
    Token number;
    while (file.read(reinterpret_cast<char *>(&number), sizeof(Token))) data->push_back(number);
    std::cout << "Loaded encoding with " << data->size() << " tokens\n";
    return true;
}

void usage() {
    std::cout << "Training, encoding and decoding of tokens\nUsage: minbpe-cc [OPTIONS]\n\nOptions:\n"
                 "  -h,--help                   Print this help message and exit\n"
                 "  -i,--input TEXT             Path to the input to be trained on, encoded or decoded\n"
                 "  -o,--output TEXT            Path for the output of the encoding or decoding\n"
                 "  -s,--special-tokens-path TEXT\n                              Path to the special tokens file\n"
                 "  -t,--train                  Train on the input\n"
                 "  -d,--decode                 Decode the input\n"
                 "  -e,--encode                 Encode the input\n"
                 "  -w,--write-vocab            When training, write the vocabulary to a file\n"
                 "  --vocab-size INT            Vocabulary size\n"
                 "  --encoder TEXT              Encoder to use from basic,gpt2,gpt4\n"
                 "  -m,--model-path TEXT        Path to load or save the model\n"
                 "  -v,--verbose                Print more things\n"
                 "  -c,--conflict-resolution TEXT:{first,lexical}\n"
                 "                              Conflict resolution strategy: 'first' or 'lexical'\n"
                 "  --device INT                HIP device used for training (default 0)\n"
                 "  --device-encode             When encoding, apply the merges on the HIP device (--device)\n";
}

}  // namespace

int main(int argc, char *argv[]) {
    std::string input_path, output_path, special_token_path, encoder = "gpt4", model_path = "./output.model";
    std::string conflict_resolution_str = "first";
    bool train = false, decode = false, encode = false, write_vocab = false, verbose = false, device_encode = false;
    int vocab_size = 512, device = 0;

    // option parsing (the reference uses CLI11, :93-133)
    for (int i = 1; i < argc; ++i) {
        std::string arg = argv[i], val;
        bool has_val = false;
        size_t eq = arg.find('=');
        if (arg.rfind("--", 0) == 0 && eq != std::string::npos) { val = arg.substr(eq + 1); arg = arg.substr(0, eq); has_val = true; }
        auto value = [&](std::string *dst) -> bool {
            if (has_val) { *dst = val; return true; }
            if (i + 1 >= argc) { std::cerr << arg << ": 1 required TEXT missing\n"; return false; }
            *dst = argv[++i];
            return true;
        };
        std::string tmp;
        if (arg == "-h" || arg == "--help") { usage(); return 0; }
        else if (arg == "-i" || arg == "--input") { if (!value(&input_path)) return 106; }
        else if (arg == "-o" || arg == "--output") { if (!value(&output_path)) return 106; }
        else if (arg == "-s" || arg == "--special-tokens-path") { if (!value(&special_token_path)) return 106; }
        else if (arg == "-m" || arg == "--model-path") { if (!value(&model_path)) return 106; }
        else if (arg == "--encoder") { if (!value(&encoder)) return 106; }
        else if (arg == "--vocab-size") {
            if (!value(&tmp)) return 106;
            try { vocab_size = std::stoi(tmp); } catch (...) { std::cerr << "--vocab-size: invalid integer " << tmp << "\n"; return 104; }
        } else if (arg == "--device") {
            if (!value(&tmp)) return 106;
            try { device = std::stoi(tmp); } catch (...) { std::cerr << "--device: invalid integer " << tmp << "\n"; return 104; }
        } else if (arg == "-c" || arg == "--conflict-resolution") {
            if (!value(&conflict_resolution_str)) return 106;
            if (conflict_resolution_str != "first" && conflict_resolution_str != "lexical") {   // CLI::IsMember, :129-131
                std::cerr << "--conflict-resolution: " << conflict_resolution_str << " not in {first,lexical}\n";
                return 105;
            }
        } else if (arg == "-t" || arg == "--train") train = true;
        else if (arg == "-d" || arg == "--decode") decode = true;
        else if (arg == "-e" || arg == "--encode") encode = true;
        else if (arg == "-w" || arg == "--write-vocab") write_vocab = true;
        else if (arg == "--device-encode") device_encode = true;
        else if (arg == "-v" || arg == "--verbose") verbose = true;
        else { std::cerr << "The following argument was not expected: " << arg << "\n"; return 109; }
    }

    if (!input_path.empty()) {                                  // :135-144
        if (!exists(input_path)) { std::cerr << "Input file " << input_path << " does not exist\n"; return -1; }
    } else {
        std::cerr << "Input file not specified\n";
        return -1;
    }

    std::string special_tokens_data;                            // :147-159
    bool have_special = false;
    if (!special_token_path.empty() && exists(special_token_path)) {
        std::string err;
        if (load_file_to_string(special_token_path, &special_tokens_data, &err)) {
            have_special = true;
            std::cout << "Loaded special tokens from " << special_token_path << "\n";
        } else {
            std::cerr << "Failed to load special tokens from " << special_token_path << ": " << err << "\n";
        }
    }

    auto t1 = std::chrono::high_resolution_clock::now();

    const char *pat = mbpe_split_pattern(encoder.c_str());       // :167-177
    if (!pat) { std::cout << "Encoder should be one of: basic, gpt2 or gpt4\n"; return -1; }

    mbpe_tokenizer *rt = nullptr;                               // Tokenizer rt(split_pattern), :179
    if (mbpe_tok_create(pat, &rt) != MBPE_OK) {
        std::cerr << "Error: " << mbpe_last_error() << "\n";
        return -1;
    }
    int rc = 0;
    if (train) {                                                // :181-211
        if (have_special) mbpe_tok_set_special_tokens(rt, special_tokens_data.data(), special_tokens_data.size());
        else std::cout << "No special tokens file provided\n";
        std::cout << "Training using file \"" << input_path << "\" encoder " << encoder << " vocab size " << vocab_size
                  << " model path " << model_path << "\n";
        if (verbose) std::cout << "Loading file " << input_path << "\n";
        std::string input, err;
        if (load_file_to_string(input_path, &input, &err)) {
            if (verbose) std::cout << "Starting training...\n";
            if (mbpe_tok_train(rt, reinterpret_cast<const uint8_t *>(input.data()), input.size(), (uint32_t)vocab_size,
                               conflict_resolution_str == "lexical" ? 1 : 0, verbose, device) != MBPE_OK) {
                std::cerr << "Error: " << mbpe_last_error() << "\n";
                rc = -1;
            } else {
                mbpe_tok_save(rt, model_path.c_str(), write_vocab);
            }
        } else {
            std::cerr << "Failed to load training input file: " << err << "\n";
        }
    } else if (encode) {                                        // :212-242
        if (output_path.empty()) { std::cerr << "Output file not specified\n"; mbpe_tok_destroy(rt); return -1; }
        if (!exists(model_path)) { std::cerr << "Model file " << model_path << " does not exist\n"; mbpe_tok_destroy(rt); return -1; }
        std::cout << "Encoding input file \"" << input_path << "\" encoder " << encoder << " model path " << model_path
                  << " output to " << output_path << "\n";
        mbpe_tok_load(rt, model_path.c_str(), verbose);
        std::string input, err;
        if (load_file_to_string(input_path, &input, &err)) {
            std::vector<Token> encoded(input.size() + 1);
            uint64_t n = 0;
            const uint8_t *in = reinterpret_cast<const uint8_t *>(input.data());
            const int erc = device_encode
                                ? mbpe_tok_encode_device(rt, in, input.size(), verbose, device, encoded.data(), encoded.size(), &n)
                                : mbpe_tok_encode(rt, in, input.size(), verbose, encoded.data(), encoded.size(), &n);
            if (erc != MBPE_OK) {
                std::cerr << "Error: " << mbpe_last_error() << "\n";
                rc = -1;
            } else {
                encoded.resize(n);
                std::cout << "Writing " << encoded.size() << " encoded tokens\n";
                if (save_encoding(output_path, encoded, &err)) std::cout << "Success\n";
                else std::cerr << "Failed with error: " << err << "\n";
            }
        } else {
            std::cerr << "Failed with error: " << err << "\n";
        }
    } else if (decode) {                                        // :243-258
        std::cout << "Decoding input file \"" << input_path << "\" encoder " << encoder << " model path " << model_path
                  << " output to " << output_path << "\n";
        mbpe_tok_load(rt, model_path.c_str(), verbose);
        std::vector<Token> input;
        std::string err;
        if (load_encoding(input_path, &input, &err)) {
            uint64_t n = 0;
            mbpe_tok_decode(rt, input.data(), input.size(), 0, nullptr, 0, &n);
            std::string decoded(n, '\0');
            mbpe_tok_decode(rt, input.data(), input.size(), verbose, reinterpret_cast<uint8_t *>(&decoded[0]), n, &n);
            std::cout << "Writing " << decoded.size() << " decoded tokens to " << output_path << "\n";
            std::ofstream file(output_path);
            if (file) file << decoded;
        } else {
            std::cerr << "Failed with error: " << err << "\n";
        }
    }
    mbpe_tok_destroy(rt);
    if (rc != 0) return rc;

    auto t2 = std::chrono::high_resolution_clock::now();       // :260-264
    auto ms_int = std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count();
    std::cout << "Execution time: " << ms_int / 1000.0 << " (s)" << std::endl;
    return 0;
}
