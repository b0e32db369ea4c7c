// placeholder until the edge-scoring kernels land (next commit)
#include "common.h"
namespace besst {
size_t score_workspace_bytes(int64_t, int64_t) { return 256; }
}
