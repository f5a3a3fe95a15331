/*
 * mbpe_tokenizer.h -- C-ABI of the host-side mirror of minbpe-cc's Tokenizer
 * (code/include/Tokenizer.h:379-927): the pieces around the GPU hot path that a
 * drop-in needs -- regex pre-split, "minbpe v1" model files, special tokens,
 * encode, decode -- plus train(), which runs the hot path through mbpe.h.
 * The C++ class behind it is minbpe-cc_amd/host/tokenizer.h (same method names
 * and argument meaning as the reference class); the `minbpe-cc` executable
 * built from host/main.cpp keeps the reference's command line.
 *
 * Return codes are mbpe_status values (mbpe.h); text of the last error:
 * mbpe_last_error().
 */
#ifndef MBPE_TOKENIZER_H
#define MBPE_TOKENIZER_H

#include "mbpe.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mbpe_tokenizer mbpe_tokenizer;

/* Tokenizer(const string &pattern), Tokenizer.h:391-451 ("" = basic). */
MBPE_API int  mbpe_tok_create(const char *pattern, mbpe_tokenizer **out);
MBPE_API void mbpe_tok_destroy(mbpe_tokenizer *t);

/* set_special_tokens_from_file, Tokenizer.h:476-486: "name id" pairs. */
MBPE_API int mbpe_tok_set_special_tokens(mbpe_tokenizer *t, const char *text, uint64_t n);

/* train, Tokenizer.h:489-598.  conflict_resolution: 1 = lexical, 0 = first
 * (the reference CLI's default); both run on HIP device `device_id` through
 * mbpe_train. */
MBPE_API int mbpe_tok_train(mbpe_tokenizer *t, const uint8_t *text, uint64_t n, uint32_t vocab_size,
                            int conflict_resolution, int verbose, int device_id);

/* Direct access to the trained / loaded merges (2 u32 per merge). */
MBPE_API int mbpe_tok_set_merges(mbpe_tokenizer *t, const uint32_t *merges, uint32_t n_merges);
MBPE_API int mbpe_tok_get_merges(mbpe_tokenizer *t, uint32_t *merges_out, uint32_t cap, uint32_t *n_out);

/* save / load, Tokenizer.h:875-926 / :754-872. */
MBPE_API int mbpe_tok_save(mbpe_tokenizer *t, const char *path, int write_vocab);
MBPE_API int mbpe_tok_load(mbpe_tokenizer *t, const char *path, int verbose);

/* encode, Tokenizer.h:653-722 (special split :605-650, regex split :664-704,
 * greedy multi-pass merge application :325-367).  tokens_out may be NULL to
 * query the count. */
MBPE_API int mbpe_tok_encode(mbpe_tokenizer *t, const uint8_t *text, uint64_t n, int verbose,
                             uint32_t *tokens_out, uint64_t cap, uint64_t *n_out);

/* The same with internal_encode (Tokenizer.h:325-377) on HIP device `device_id`
 * (mbpe_encode_chunks); special-token and regex splitting stay on the host.  No CPU
 * fallback: MBPE_ERR_NO_DEVICE without a device. */
MBPE_API int mbpe_tok_encode_device(mbpe_tokenizer *t, const uint8_t *text, uint64_t n, int verbose, int device_id,
                                    uint32_t *tokens_out, uint64_t cap, uint64_t *n_out);

/* decode, Tokenizer.h:725-751.  bytes_out may be NULL to query the length. */
MBPE_API int mbpe_tok_decode(mbpe_tokenizer *t, const uint32_t *tokens, uint64_t n, int verbose,
                             uint8_t *bytes_out, uint64_t cap, uint64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif
