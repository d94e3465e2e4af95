// q3_codec.cpp — 12 Hz codec decoder orchestration (placeholder until the kernels land)
#include "q3_engine.h"
namespace q3 {
struct CodecW {};
void Engine::codec_finalize() {}
void Engine::codec_free() {}
int64_t Engine::codec_run(const int32_t*, int, float**) { throw Error("codec decoder not built yet"); }
}
