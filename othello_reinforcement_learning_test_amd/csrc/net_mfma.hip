// net_mfma.hip -- placeholder until the MFMA trunk lands (next commit)
#include "net.h"
namespace oth {
struct MfmaWeights { int dummy; };
int mfma_pack_weights(oth_net*, int) { set_error("MFMA trunk not built yet"); return OTH_E_UNSUPPORTED; }
void mfma_free_weights(oth_net* net) { delete net->mfma; net->mfma = nullptr; }
int mfma_forward(oth_net*, const uint64_t*, const uint64_t*, const uint64_t*, int64_t, const int32_t*, float*, float*, hipStream_t) {
    set_error("MFMA trunk not built yet"); return OTH_E_UNSUPPORTED; }
}
