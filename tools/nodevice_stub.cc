// Sanitizer harness only (tools/asan_host_check.sh): stands in for csrc/rl_render.hip so that the HOST side of the
// library (ABI, OBJ/MTL loader, BVH builder, codecs, registries) can be built with g++ -fsanitize=address,undefined
// and run through tests/test_host_logic.py.  Never part of libraylib.so.
#include "rl_host.h"
namespace rl {
bool DeviceAvailable() { return false; }
bool DeviceRender(Scene&, const RenderRequest&, RaylibAMDStats&) { return false; }
bool DeviceClosestHit(Scene&, const float*, int32_t, float, void*) { return false; }
bool DevicePostProcess(Image&) { return false; }
void* DeviceImagePixels(Image&) { return nullptr; }
bool DeviceReadback(Image&) { return false; }
bool DeviceDumpRGB(Image&, float*) { return false; }
void DeviceFreePixels(void*) {}
bool DeviceEvalMath(int, const float*, const float*, int, float*) { return false; }
bool DeviceEvalHook(int, Scene*, const DCamera*, int, int, const float*, int, uint64_t, float*) { return false; }
void DeviceReleaseScene(DeviceScene*) {}
void DeviceShutdown() {}
int DeviceNumRanks() { return 0; }
bool DeviceDrain(RaylibAMDStats*) { return false; }
bool DeviceVerifyExactMath(int, uint64_t*, uint64_t*) { return false; }
}
