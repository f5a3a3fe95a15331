// Encode on the device: internal_encode / internal_internal_encode (reference Tokenizer.h:325-377)
// for all chunks of a text at once.
//
// The reference encodes a chunk with repeated left-to-right passes; a pass replaces ANY pair found in
// merges_lookup (first match wins, not rank order: `i += 2` after a hit) and the passes stop when one
// makes no replacement (:325-367).  A pass looks sequential but is not: call position i a CANDIDATE when
// (t[i], t[i+1]) is in merges_lookup and both tokens belong to the same chunk, and let r[i] be the number
// of consecutive candidates immediately before i.  The walk replaces at i  <=>  i is a candidate and r[i]
// is even, and it drops t[i]  <=>  r[i] is odd (then i-1 was replaced and swallowed it).  r[i] is a
// segmented count, so a pass is: look up every pair -> parity of r by a scan -> compact.  Chunks never
// interact (a candidate never starts at the last token of a chunk), a chunk that no longer changes stays
// as it is, so passes over the whole text until nothing changes give every chunk its reference result.
//
// Layout: 32-bit tokens (the reference's Token, Tokenizer.h:37), bit 31 = "last token of its chunk".
// A span = 1,024 consecutive tokens = the unit one wave walks (16 x 64 lanes, ballots give the
// candidate masks); spans are linked by two small scans (run parity, output offsets).
#include "mbpe.h"
#include "../host/mbpe_host.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

constexpr uint32_t kEnd = 0x80000000u;     // last token of its chunk
constexpr uint32_t kNone = 0xFFFFFFFFu;    // no candidate / token removed
constexpr uint32_t kDrop = 0x7FFFFFFEu;    // byte of a chunk that collapses to one token (Tokenizer.h:86-93)
constexpr uint32_t kIdMask = 0x7FFFFFFFu;
constexpr int kWave = 64;
constexpr int kSpan = 1024;
constexpr int kSpanIters = kSpan / kWave;
constexpr int kEncThreads = 256;           // 4 waves = 4 spans per workgroup
constexpr unsigned long long kEmptyKey = ~0ull;

struct EncLut {
    const unsigned long long *keys;        // (first << 32) | second, kEmptyKey when free
    const uint32_t *vals;                  // id of the merged token
    uint32_t shift;                        // 64 - log2(capacity)
    uint32_t mask;
};

__host__ __device__ inline uint32_t enc_hash(unsigned long long key, uint32_t shift) {
    return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> shift);
}

__device__ __forceinline__ uint32_t enc_lookup(const EncLut &lut, uint32_t a, uint32_t b) {
    const unsigned long long key = ((unsigned long long)a << 32) | b;
    uint32_t h = enc_hash(key, lut.shift);
    for (;;) {
        const unsigned long long k = lut.keys[h];
        if (k == key) return lut.vals[h];
        if (k == kEmptyKey) return kNone;
        h = (h + 1) & lut.mask;
    }
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (kWave - 1); }

// text_to_vector, Tokenizer.h:94-99 (char_to_token :80-82) + chunk ends
__global__ void k_enc_widen(const uint8_t *__restrict__ text, uint64_t n, const uint8_t *__restrict__ endmask,
                            uint32_t *__restrict__ tok) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t e = (endmask[i >> 3] >> (i & 7)) & 1u;
        tok[i] = text[i] | (e ? kEnd : 0u);
    }
}

// chunks that are ONE token (NUL + a number: how encode passes special tokens, Tokenizer.h:635-637, :667-670)
struct SingleChunk { unsigned long long start, len; uint32_t id, pad; };
__global__ void k_enc_single(const SingleChunk *__restrict__ sc, uint32_t n_sc, uint32_t *__restrict__ tok) {
    const uint32_t c = blockIdx.x;
    if (c >= n_sc) return;
    const SingleChunk s = sc[c];
    for (unsigned long long i = threadIdx.x; i < s.len; i += blockDim.x)
        tok[s.start + i] = i == 0 ? (s.id | kEnd) : kDrop;
}

// pass, step 1: cand[i] = id of the token that (t[i], t[i+1]) merges to, or kNone; per span: are all its
// positions candidates, and the parity of its trailing run of candidates
__global__ __launch_bounds__(kEncThreads) void k_enc_cand(const uint32_t *__restrict__ tok, uint64_t n, EncLut lut,
                                                          uint32_t *__restrict__ cand,
                                                          uint32_t *__restrict__ span_sum) {
    const uint64_t span = (uint64_t)blockIdx.x * (kEncThreads / kWave) + threadIdx.x / kWave;
    const uint64_t base = span * kSpan;
    if (base >= n) return;
    const uint32_t lane = lane_id();
    bool all = true;
    uint32_t par = 0;
    for (int it = 0; it < kSpanIters; ++it) {
        const uint64_t i = base + (uint64_t)it * kWave + lane;
        const uint32_t t = i < n ? tok[i] : kDrop;
        uint32_t nx = __shfl_down(t, 1, kWave);
        if (lane == kWave - 1) nx = i + 1 < n ? tok[i + 1] : kDrop;
        uint32_t c = kNone;
        if (i < n && !(t & kEnd) && t != kDrop && nx != kDrop) c = enc_lookup(lut, t, nx & kIdMask);
        if (i < n) cand[i] = c;
        const unsigned long long M = __ballot(c != kNone);
        if (M != ~0ull) {
            all = false;
            par = (uint32_t)__builtin_clzll(~M) & 1u;        // candidates at the top of this group
        }                                                   // (a full group adds 64: parity unchanged)
    }
    if (lane == 0) span_sum[span] = (all ? 1u : 0u) | (par << 1);
}

// One workgroup, two sweeps: out[s] = the fold of elements 0 .. s-1 under an associative operator.
// Thread t owns a contiguous slice of the spans.
//   parity scan: element (all, par); L then R = R.all ? (L.all, L.par ^ R.par) : R      [a full span has
//                                                                                        even length]
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void k_enc_scan_parity(const uint32_t *__restrict__ span_sum,
                                                                  uint64_t n_spans, uint32_t *__restrict__ in_par) {
    __shared__ uint32_t sh[kScanThreads];
    const uint64_t per = (n_spans + kScanThreads - 1) / kScanThreads;
    const uint64_t lo = per * threadIdx.x, hi = lo + per < n_spans ? lo + per : n_spans;
    uint32_t all = 1, par = 0;
    for (uint64_t s = lo; s < hi; ++s) {
        const uint32_t v = span_sum[s];
        if (v & 1u) par ^= v >> 1; else { all = 0; par = v >> 1; }
    }
    sh[threadIdx.x] = all | (par << 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 1, p = 0;                 // nothing before the text: an empty (full, even) prefix
        for (int t = 0; t < kScanThreads; ++t) {
            const uint32_t v = sh[t];
            sh[t] = p;                         // parity of the run of candidates right before slice t
            if (v & 1u) p ^= v >> 1; else { a = 0; p = v >> 1; }
        }
        (void)a;
    }
    __syncthreads();
    par = sh[threadIdx.x];
    for (uint64_t s = lo; s < hi; ++s) {
        in_par[s] = par;
        const uint32_t v = span_sum[s];
        if (v & 1u) par ^= v >> 1; else par = v >> 1;
    }
}

//   sum scan: exclusive prefix sums of the spans' kept-token counts; total to *total
__global__ __launch_bounds__(kScanThreads) void k_enc_scan_sum(const uint32_t *__restrict__ cnt, uint64_t n_spans,
                                                               unsigned long long *__restrict__ off,
                                                               unsigned long long *__restrict__ total) {
    __shared__ unsigned long long sh[kScanThreads];
    const uint64_t per = (n_spans + kScanThreads - 1) / kScanThreads;
    const uint64_t lo = per * threadIdx.x, hi = lo + per < n_spans ? lo + per : n_spans;
    unsigned long long s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += cnt[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long acc = 0;
        for (int t = 0; t < kScanThreads; ++t) { const unsigned long long v = sh[t]; sh[t] = acc; acc += v; }
        *total = acc;
    }
    __syncthreads();
    s = sh[threadIdx.x];
    for (uint64_t i = lo; i < hi; ++i) { off[i] = s; s += cnt[i]; }
}

// pass, step 2: with the parity of the candidate run before every position, decide.  cand[i] becomes the
// value position i contributes to the next stream (kNone: nothing).
__global__ __launch_bounds__(kEncThreads) void k_enc_match(const uint32_t *__restrict__ tok, uint64_t n,
                                                           uint32_t *__restrict__ cand,
                                                           const uint32_t *__restrict__ in_par,
                                                           uint32_t *__restrict__ span_keep) {
    const uint64_t span = (uint64_t)blockIdx.x * (kEncThreads / kWave) + threadIdx.x / kWave;
    const uint64_t base = span * kSpan;
    if (base >= n) return;
    const uint32_t lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t carry = in_par[span];             // parity of the run of candidates right before this group
    uint32_t kept = 0;
    for (int it = 0; it < kSpanIters; ++it) {
        const uint64_t i = base + (uint64_t)it * kWave + lane;
        const uint32_t c = i < n ? cand[i] : kNone;
        const uint32_t t = i < n ? tok[i] : kDrop;
        uint32_t nx = __shfl_down(t, 1, kWave);
        if (lane == kWave - 1) nx = i + 1 < n ? tok[i + 1] : 0u;
        const unsigned long long M = __ballot(c != kNone);
        // r = consecutive candidates immediately below this lane (continuing into `carry` if all of them are)
        const unsigned long long zeros_below = ~M & lt;
        uint32_t r;
        if (zeros_below == 0ull) r = lane + carry;
        else r = lane - 1u - (63u - (uint32_t)__builtin_clzll(zeros_below));
        const bool odd = r & 1u;
        uint32_t v = kNone;
        if (i < n && !odd && t != kDrop) v = c != kNone ? (c | (nx & kEnd)) : t;   // replaced, or kept as it is
        if (i < n) cand[i] = v;
        kept += (uint32_t)__popcll(__ballot(v != kNone));
        if (M != ~0ull) carry = (uint32_t)__builtin_clzll(~M) & 1u;
    }
    if (lane == 0) span_keep[span] = kept;
}

// pass, step 3: compact
__global__ __launch_bounds__(kEncThreads) void k_enc_scatter(const uint32_t *__restrict__ val, uint64_t n,
                                                             const unsigned long long *__restrict__ span_off,
                                                             uint32_t *__restrict__ out) {
    const uint64_t span = (uint64_t)blockIdx.x * (kEncThreads / kWave) + threadIdx.x / kWave;
    const uint64_t base = span * kSpan;
    if (base >= n) return;
    const uint32_t lane = lane_id();
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long o = span_off[span];
    for (int it = 0; it < kSpanIters; ++it) {
        const uint64_t i = base + (uint64_t)it * kWave + lane;
        const uint32_t v = i < n ? val[i] : kNone;
        const unsigned long long K = __ballot(v != kNone);
        if (v != kNone) out[o + (uint32_t)__popcll(K & lt)] = v;
        o += (uint32_t)__popcll(K);
    }
}

std::string hip_err(const char *what, hipError_t e) { return std::string(what) + ": " + hipGetErrorString(e); }

#define HIPCHK(expr)                                                   \
    do {                                                               \
        hipError_t e__ = (expr);                                       \
        if (e__ != hipSuccess) {                                       \
            mbpe_host::set_last_error(hip_err(#expr, e__));            \
            rc = e__ == hipErrorOutOfMemory ? MBPE_ERR_OOM : MBPE_ERR_HIP; \
            goto done;                                                 \
        }                                                              \
    } while (0)

// std::stoi on the remainder of a NUL-led chunk (Tokenizer.h:86-93): value when it parses
bool stoi_value(const uint8_t *s, uint64_t n, long long *out) {
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
    if (v > 2147483647LL || v < -2147483648LL) return false;
    *out = v;
    return true;
}

}  // namespace

extern "C" int mbpe_encode_chunks(int device_id, const uint8_t *text, uint64_t n_bytes, const uint64_t *chunk_off,
                                  uint64_t n_chunks, const uint32_t *merges, uint32_t n_merges, uint32_t *tokens_out,
                                  uint64_t cap, uint64_t *n_out, uint32_t *n_passes_out) {
    if (!n_out || (!text && n_bytes) || (!merges && n_merges)) {
        mbpe_host::set_last_error("mbpe_encode_chunks: NULL argument");
        return MBPE_ERR_ARG;
    }
    *n_out = 0;
    if (n_passes_out) *n_passes_out = 0;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device_id < 0 || device_id >= n_dev) {
        mbpe_host::set_last_error("no usable HIP device (the MI355X path has no CPU fallback)");
        return MBPE_ERR_NO_DEVICE;
    }
    const uint64_t one[2] = {0, n_bytes};
    if (!chunk_off) { chunk_off = one; n_chunks = 1; }
    if (chunk_off[0] != 0 || chunk_off[n_chunks] != n_bytes) {
        mbpe_host::set_last_error("chunk_off must start at 0 and end at n_bytes");
        return MBPE_ERR_ARG;
    }
    if (n_bytes == 0) return MBPE_OK;

    // host side: chunk-end bits, chunks that are one token, the pair -> id table
    std::vector<uint8_t> mask((n_bytes + 7) / 8 + 8, 0);
    std::vector<SingleChunk> singles;
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const uint64_t s = chunk_off[c], e = chunk_off[c + 1];
        if (e < s) { mbpe_host::set_last_error("chunk_off must be ascending"); return MBPE_ERR_ARG; }
        if (e == s) continue;
        mask[(e - 1) >> 3] |= (uint8_t)(1u << ((e - 1) & 7));
        long long id;
        if (text[s] == 0 && stoi_value(text + s + 1, e - s - 1, &id)) {
            if ((uint32_t)(int)id >= kDrop) {
                mbpe_host::set_last_error("token id of a NUL-led chunk does not fit 31 bits");
                return MBPE_ERR_ARG;
            }
            singles.push_back({s, e - s, (uint32_t)(int)id, 0});
        }
    }
    uint32_t bits = 4;
    while ((1ull << bits) < 2ull * n_merges + 2) ++bits;
    const uint32_t capacity = 1u << bits;
    std::vector<unsigned long long> keys(capacity, kEmptyKey);
    std::vector<uint32_t> vals(capacity, 0);
    for (uint32_t k = 0; k < n_merges; ++k) {       // merges_lookup[pair] = 256 + k: a repeated pair keeps the last id
        const unsigned long long key = ((unsigned long long)merges[2 * k] << 32) | merges[2 * k + 1];
        uint32_t h = enc_hash(key, 64 - bits);
        while (keys[h] != kEmptyKey && keys[h] != key) h = (h + 1) & (capacity - 1);
        keys[h] = key;
        vals[h] = 256 + k;
    }

    int rc = MBPE_OK;
    uint8_t *d_text = nullptr, *d_mask = nullptr;
    uint32_t *tok[2] = {nullptr, nullptr}, *cand = nullptr, *span_a = nullptr, *span_b = nullptr, *d_vals = nullptr;
    unsigned long long *span_off = nullptr, *d_total = nullptr, *d_keys = nullptr;
    SingleChunk *d_singles = nullptr;
    hipStream_t stream = nullptr;
    uint64_t n = n_bytes;
    int cur = 0;
    uint32_t passes = 0;
    {
        const uint64_t n_spans0 = (n + kSpan - 1) / kSpan;
        HIPCHK(hipSetDevice(device_id));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIPCHK(hipMalloc(&d_text, n_bytes));
        HIPCHK(hipMalloc(&d_mask, mask.size()));
        HIPCHK(hipMalloc(&tok[0], n * 4));
        HIPCHK(hipMalloc(&tok[1], n * 4));
        HIPCHK(hipMalloc(&cand, n * 4));
        HIPCHK(hipMalloc(&span_a, n_spans0 * 4));
        HIPCHK(hipMalloc(&span_b, n_spans0 * 4));
        HIPCHK(hipMalloc(&span_off, n_spans0 * 8));
        HIPCHK(hipMalloc(&d_total, 8));
        HIPCHK(hipMalloc(&d_keys, (size_t)capacity * 8));
        HIPCHK(hipMalloc(&d_vals, (size_t)capacity * 4));
        HIPCHK(hipMemcpyAsync(d_text, text, n_bytes, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(d_keys, keys.data(), (size_t)capacity * 8, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(d_vals, vals.data(), (size_t)capacity * 4, hipMemcpyHostToDevice, stream));
        const int wblocks = (int)std::min<uint64_t>((n + 255) / 256, 8192);
        hipLaunchKernelGGL(k_enc_widen, dim3(wblocks), dim3(256), 0, stream, d_text, n, d_mask, tok[0]);
        if (!singles.empty()) {
            HIPCHK(hipMalloc(&d_singles, singles.size() * sizeof(SingleChunk)));
            HIPCHK(hipMemcpyAsync(d_singles, singles.data(), singles.size() * sizeof(SingleChunk),
                                  hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(k_enc_single, dim3((uint32_t)singles.size()), dim3(64), 0, stream, d_singles,
                               (uint32_t)singles.size(), tok[0]);
        }
        const EncLut lut = {d_keys, d_vals, 64 - bits, capacity - 1};
        for (;;) {
            const uint64_t n_spans = (n + kSpan - 1) / kSpan;
            const dim3 grid((uint32_t)((n_spans + kEncThreads / kWave - 1) / (kEncThreads / kWave))), block(kEncThreads);
            hipLaunchKernelGGL(k_enc_cand, grid, block, 0, stream, tok[cur], n, lut, cand, span_a);
            hipLaunchKernelGGL(k_enc_scan_parity, dim3(1), dim3(kScanThreads), 0, stream, span_a, n_spans, span_b);
            hipLaunchKernelGGL(k_enc_match, grid, block, 0, stream, tok[cur], n, cand, span_b, span_a);
            hipLaunchKernelGGL(k_enc_scan_sum, dim3(1), dim3(kScanThreads), 0, stream, span_a, n_spans, span_off, d_total);
            hipLaunchKernelGGL(k_enc_scatter, grid, block, 0, stream, cand, n, span_off, tok[1 - cur]);
            unsigned long long total = 0;
            HIPCHK(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipGetLastError());
            ++passes;
            if (total == n) break;             // the pass changed nothing: tok[cur] is the result (so is tok[1 - cur])
            n = total;
            cur = 1 - cur;
            if (n == 0) break;
        }
        *n_out = n;
        if (n_passes_out) *n_passes_out = passes;
        if (tokens_out) {
            if (cap < n) { mbpe_host::set_last_error("tokens_out too small"); rc = MBPE_ERR_ARG; goto done; }
            if (n) HIPCHK(hipMemcpy(tokens_out, tok[cur], n * 4, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < n; ++i) tokens_out[i] &= kIdMask;      // strip the chunk-end flags
        }
    }
done:
    (void)hipFree(d_text); (void)hipFree(d_mask); (void)hipFree(tok[0]); (void)hipFree(tok[1]); (void)hipFree(cand);
    (void)hipFree(span_a); (void)hipFree(span_b); (void)hipFree(span_off); (void)hipFree(d_total);
    (void)hipFree(d_keys); (void)hipFree(d_vals); (void)hipFree(d_singles);
    if (stream) (void)hipStreamDestroy(stream);
    return rc;
}
