// swapnet_amd -- texture-stage model (placeholder until the texture builders land).
#include "engine.h"
namespace swn {
Model* create_texture_model(Ctx&, int, int, int, bool, int) { throw Error(1, "texture model: not built yet"); }
}  // namespace swn
