// pb_embed.hip -- embed half of the C ABI (placeholder until the EfficientNet kernels land: fails loudly).
#include "pb_common.h"

struct pb_embedder { int unused; };

extern "C" {
int pb_embed_create(pb_embedder **out, int, const void *, size_t, uint32_t) {
    if (out) *out = nullptr;
    return pb::fail(PB_ERR_INTERNAL, "pb_embed_create: embed kernels not built in this revision");
}
int pb_embed_destroy(pb_embedder *) { return PB_OK; }
int pb_embed_info(const pb_embedder *, uint32_t *, uint32_t *, uint32_t *, uint32_t *) { return pb::fail(PB_ERR_INTERNAL, "not built"); }
int pb_embed_batch(pb_embedder *, const uint8_t *, uint32_t, uint8_t *, float *) { return pb::fail(PB_ERR_INTERNAL, "not built"); }
int pb_embed_batch_device(pb_embedder *, const uint8_t *, uint32_t, uint8_t *, float *) { return pb::fail(PB_ERR_INTERNAL, "not built"); }
int pb_mlhash(pb_embedder *, const uint8_t *, uint8_t *, size_t) { return pb::fail(PB_ERR_INTERNAL, "not built"); }
int pb_embed_set_option(pb_embedder *, int, int64_t) { return pb::fail(PB_ERR_INTERNAL, "not built"); }
}
