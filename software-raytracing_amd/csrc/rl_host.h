// Host-side object model of the MI355X raylib (everything behind the C-ABI that is
// not a kernel): OBJ models, images, cameras, scenes, and the flattening step
// that turns them into the device layout of rl_device.h.
#pragma once

#include "raylib_types.h"
#include "raylib_amd.h"
#include "rl_device.h"

#include <math.h>
#include <stdint.h>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace rl {

// ---------------------------------------------------------------------------
// float3 with the arithmetic conventions the reference's results depend on
// (reference core/vec3.h): normalize multiplies by 1/length, dot sums left to
// right, cross negates the middle term.  Host code is compiled with
// -ffp-contract=off so that these produce the same bits as the reference build.
struct f3 { float x, y, z; };
inline f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
inline f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline f3 operator*(f3 a, float t) { return F3(a.x * t, a.y * t, a.z * t); }
inline f3 operator*(float t, f3 a) { return F3(a.x * t, a.y * t, a.z * t); }
inline f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline f3 cross(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x); }
inline float length(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
inline f3 normalize(f3 a) { float k = 1.0f / length(a); return F3(a.x * k, a.y * k, a.z * k); }
// std::min / std::max selection order (reference core/vec3.h:151-162): min(a,b) = (b < a) ? b : a
inline f3 fmin3(f3 a, f3 b) { return F3(b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z); }
inline f3 fmax3(f3 a, f3 b) { return F3(a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z); }

// ---------------------------------------------------------------------------
// 26-word triangle / 19-word material records: the host-side truth, identical in
// layout to what RaylibAMD_SceneExport* hand out.
struct HostTriangle {
	f3 v0, v1, v2;
	f3 n0, n1, n2;
	float s0, t0, s1, t1, s2, t2;
	int32_t material;
	int32_t shape;
};
static_assert(sizeof(HostTriangle) == 104, "HostTriangle layout");

enum MaterialType { MAT_LAMBERTIAN = 0, MAT_MIRROR = 1, MAT_DIELECTRIC = 2, MAT_MICROFACET = 3, MAT_METAL = 4, MAT_DIFFUSE_LIGHT = 5 };

struct HostMaterial {
	int32_t type;
	float albedo[3];
	float roughness, metallic;
	float emissive[3];
	float ior;
	float transmission[3];
	float fuzziness;
	int32_t tex[5];   // albedo, normal, roughness, metallic, emissive; -1 = none
};
static_assert(sizeof(HostMaterial) == 76, "HostMaterial layout");

uint64_t NextImageVersion();   // rl_image_io.cc: 1, 2, 3, ... over all images of the process
struct Image {
	uint32_t width = 0, height = 0;
	std::vector<float> rgba;          // 4 floats per pixel, row 0 = top
	// device-resident copy (float4 per pixel): written by Raylib_Render, consumed by Raylib_PostProcess
	void* devPixels = nullptr;
	size_t devBytes = 0;
	bool devValid = false;
	// Raylib_Render / Raylib_PostProcess leave the result on the device and mark the host pixels stale; whoever reads `rgba`
	// on the host calls SyncHost() first (one read-back when the pixels are asked for, none per render)
	bool hostStale = false;
	// bumped whenever the pixels change (reallocation, a render into the image, PostProcess): a scene that uses the image as its
	// sky panorama re-reads it at the next render, as the reference does through the handle (renderer.cc:159-176)
	// The value is unique in the PROCESS, not per image (NextImageVersion): a new image that malloc puts at a destroyed image's address
	// can never look like that image to a cache keyed on (pointer, version) -- the device copy of the sky, rl_runtime.inl SyncSky.
	uint64_t version = NextImageVersion();
	void Touch() { version = NextImageVersion(); }
	void SyncHost() const;
	Image() = default;
	Image(const Image& o) : width(o.width), height(o.height) { o.SyncHost(); rgba = o.rgba; }
	Image& operator=(const Image& o) { o.SyncHost(); width = o.width; height = o.height; rgba = o.rgba; devValid = false; hostStale = false; Touch(); return *this; }
	~Image();
	void Reallocate(uint32_t w, uint32_t h, float r, float g, float b, float a);
};

struct OBJModel {
	std::vector<HostTriangle> triangles;
	std::vector<HostMaterial> materials;           // MTL order, then the fallback Lambertian(0.5)
	std::vector<std::string> materialNames;
	std::vector<std::shared_ptr<Image>> images;    // textures referenced by materials[].tex
	int32_t numShapes = 0;
	bool finalized = false;
};

struct Camera {
	// settable state (reference render/camera.h:80-93)
	f3 origin = F3(0, 0, 0), lookAt = F3(0, 0, -1);
	float fovY_degrees = 60.0f, aspectWH = 16.0f / 9.0f;
	float aperture = 0.0f, focalDistance = 1.0f;
	float beginTime = 0.0f, endTime = 0.0f;
	// derived (reference render/camera.h:55-78)
	float lensRadius, timePeriod;
	f3 top_left, horizontal, vertical, u, v, w;
	void UpdateInternal();
	DCamera ToDevice() const;
};

struct BVH {
	std::vector<DNode> nodes;
	std::vector<DNode4> nodes4;       // the same tree collapsed to <= 4 children per node (empty for small scenes / analytic primitives)
	std::vector<DNode4Q> nodes4q;     // nodes4 on the 8-bit grid (same indices, same children): what the kernels walk by default
	uint32_t stackNeed4 = 0;          // worst-case traversal-stack entries of nodes4 (sum of children - 1 along the deepest path)
	std::vector<DNode8> nodes8;       // the 8-wide tree (DNode8, rl_device.h); its leaves are the BVH2's leaves, and the order of the triangle slots is ITS order (a node's leaf children hold consecutive slots)
	uint32_t depth8 = 0;              // levels of nodes8: a traversal's stack holds at most one entry (the rest of a node's hit children) per level
	float sahNodes4 = 0.0f, sahNodes8 = 0.0f;   // sum over the wide tree's nodes of (node's surface area / root's): the expected node steps of a random ray through either tree
	std::vector<DNode4> leafList;     // scenes of <= 4 * RL_LEAFLIST_RECORDS leaves: every leaf's box, four to a record, no inner nodes (k_trace's flat walk; empty otherwise)
	std::vector<uint32_t> triOrder;   // leaf order -> index into the flat triangle array
	uint32_t depth = 0;
	float sahCost = 0.0f;
};
// Binned-SAH BVH2 over primitive bounds.  A leaf holds <= 4 triangles or one analytic primitive.
enum PrimKind { PRIM_TRIANGLE = 0, PRIM_SPHERE = 1, PRIM_CUBE = 2 };
struct PrimRef { f3 mn, mx; uint8_t kind; uint32_t index; };
// wideGreedy: collapse the 8-wide tree by opening the largest child first instead of the plan that minimises the summed node area (an A/B switch: the scene reads
// RAYLIB_WIDE_GREEDY once, when it is finalized).  The 8-wide collapse decides the ORDER OF THE TRIANGLE SLOTS, and with it which of two surfaces at exactly the
// same distance wins (the lower slot: DESIGN.md section 4) -- frames of the two settings may differ in such tie pixels.
// splitLeaves8: give every leaf of a scene whose rays are expected to take >= minSteps8 node steps on the 4-wide tree (the scenes that walk the 8-wide tree by
// default) a split of its triangles for the 8-wide plan to open in otherwise empty slots (rl_bvh.cc SplitLeaves); triCost8: what an expected triangle test
// weighs against an expected node step in that plan.
// (Round 5, measured: triangle tests per ray 2.26 -> 1.96 ... 2.03 on the 298 k room whatever triCost8 between 0.25 and 2, no frame gets shorter -- the triangle steps
//  are a twelfth of the walk --, and the binary and 4-wide trees, emitted from the same nodes, gain a level or two: off unless RAYLIB_W8_SPLIT=1 asks for it.)
struct BVHBuildOptions { bool wideGreedy = false; bool splitLeaves8 = false; float triCost8 = 0.5f; float minSteps8 = 40.0f; };
void BuildBVH(const std::vector<PrimRef>& prims, BVH& out, const BVHBuildOptions& opt = BVHBuildOptions());
bool BVHCapacityOk(size_t numPrimitives);   // a leaf reference addresses 2^25 primitive slots
bool ValidateBVH(const BVH& bvh, const std::vector<HostTriangle>& tris);
bool ValidateBVH4(const BVH& bvh, const std::vector<HostTriangle>& tris);
bool ValidateBVH8(const BVH& bvh, const std::vector<HostTriangle>& tris);   // true also when the scene has no 8-wide tree
// the megakernel's 8-wide walk restated on the host (box arithmetic, visiting order, groups): least distance among the triangles of the leaf children it reaches
bool Walk8Host(const BVH& bvh, const std::vector<HostTriangle>& tris, const float* rays, int n, float tMin, const float* tMax, float* outT, uint32_t* outSteps);

// Scene elements created through include/raylib_amd.h (the reference's procedural scenes `new` C++ objects in the
// application instead: src/main.cc:913-984).  A material is owned by the library and shared by reference.
struct MaterialObj { HostMaterial m; };
struct SceneElement {
	PrimKind kind;
	MaterialObj* material;
	// sphere (reference geom/sphere.h:22-25)
	f3 center; float radius;
	// cube (reference geom/cube.h:36-41)
	f3 minBounds, maxBounds; float timeStartMove; f3 velocity;
	// loose triangle (reference geom/triangle.h)
	HostTriangle tri;
};
struct HostSphere { f3 center; float radius; int32_t material; };
struct HostCube { f3 minBounds, maxBounds; float timeStartMove; f3 velocity; int32_t material; };

struct DeviceScene;   // rl_render.hip

struct Scene {
	std::vector<OBJModel*> models;    // borrowed (reference raylib.cc:264-268)
	std::vector<SceneElement*> elements;   // borrowed (reference raylib.cc:258-262)
	std::vector<HostSphere> spheres;  // flattened at Finalize
	std::vector<HostCube> cubes;
	Image* sky = nullptr;             // borrowed (reference raylib.cc:270-273)
	f3 sunIlluminance = F3(0, 0, 0);
	f3 sunDirection;
	bool finalized = false;

	// flattened at Raylib_FinalizeScene
	std::vector<HostTriangle> triangles;
	std::vector<HostMaterial> materials;
	std::vector<std::shared_ptr<Image>> textures;
	BVH bvh;
	DeviceScene* device = nullptr;    // uploaded lazily at first render

	bool hasMovingCubes = false;
	float accelT0 = 0.0f, accelT1 = 0.0f;   // shutter interval the BVH boxes cover

	Scene();
	~Scene();
	void Finalize();
	bool BuildAccel(float t0, float t1);   // false: the scene exceeds the BVH's addressing (logged); nothing was built
};

// loaders (rl_obj_loader.cc, rl_image_io.cc)
bool LoadOBJ(const char* path, OBJModel& out);
void TransformOBJ(OBJModel& m, float tx, float ty, float tz, float yaw, float pitch, float roll, float sx, float sy, float sz);
Image* LoadImageFile(const char* path);
bool WriteImageFile(const Image& img, const char* path, uint32_t fileType);
void PostProcessHost(Image& img);   // see rl_abi.cc: used only when the image never lived on a device

// logging (rl_log.cc)
void Log(const char* fmt, ...);
void LogStart();
void LogFlush();
void LogStop();

// device side (rl_render.hip)
struct RenderRequest {
	RendererSettings settings;
	DCamera camera;
	uint64_t seed;
	uint32_t cellFirst, cellStride;
	bool cellMajor = false; // the rank's cells back to back even when it owns every cell (one rank going through the gather path)
	int slot = 0;           // frame slot (0 / 1) of the rank's events and counter buffers: a multi-rank frame in flight while the next is enqueued
	void* outDevice;        // may be null
	bool callerOwnsOut = false; // outDevice is memory of the CALLER (RaylibAMD_RenderDevice), not an image of the library: the call returns with the frame complete in it
	float* outHostRGBA;     // may be null; receives what outDevice would (row-major image or the rank's cells)
};
bool DeviceAvailable();
int DeviceNumRanks();                          // logical ranks (RAYLIB_NUM_GPUS) the library drives from this process; 0 without a device
float ParseDecimalFloat(const char* token);   // csrc/rl_obj_loader.cc: the OBJ parser's number reader (= strtof, with exact fast paths)
// Decoders refuse images beyond this many pixels (16384 x 16384), and images whose claimed size the file cannot plausibly hold:
// a corrupt header must not turn into a multi-gigabyte allocation.
constexpr uint64_t kMaxImagePixels = 1ull << 28;
inline bool PlausibleImageSize(uint64_t w, uint64_t h, uint64_t fileBytes, uint64_t maxPixelsPerByte)
{
	return w > 0 && h > 0 && w <= 65535 && h <= 65535 && w * h <= kMaxImagePixels && w * h <= (fileBytes + 64) * maxPixelsPerByte;
}
// csrc/rl_jpeg.cc: top-down RGBA8
bool DecodeJPEG(const std::vector<uint8_t>& data, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba);
bool DecodeTGA(const std::vector<uint8_t>& data, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba);
bool EncodeJPEG(uint32_t w, uint32_t h, const uint8_t* rgbTopDown, std::vector<uint8_t>& out);   // baseline, quality 75, 4:2:0
bool DeviceRender(Scene& scene, const RenderRequest& req, RaylibAMDStats& stats);
bool DeviceDrain(RaylibAMDStats* outLastStats);   // waits for multi-rank frames in flight; true + stats when that completed the last render call's numbers
bool DeviceClosestHit(Scene& scene, const float* rays, int32_t n, float tMin, void* outHits);
bool DevicePostProcess(Image& img);          // Image2D::PostProcess on the device; false when no device
bool DeviceDumpRGB(Image& img, float* outRGB);   // a device-resident frame packed to RGB on the device and copied to caller memory through pinned staging; false: not applicable
bool DeviceReadback(Image& img);            // device copy -> img.rgba (the caller checked hostStale)
void* DeviceImagePixels(Image& img);          // (re)allocates img.devPixels for width*height float4; nullptr when no device
void DeviceFreePixels(void* p);
// cells outside the scene's silhouette (rl_cull.cc)
struct CullScene { double boundsMin[3] = { 0, 0, 0 }, boundsMax[3] = { 0, 0, 0 }; bool boundsValid = false, prims = false, hasSky = false, hasSun = false; float sunDirection[3] = { 0, 0, 0 }, sunIlluminance[3] = { 0, 0, 0 }; };
struct CullResult { std::vector<uint32_t> active; std::vector<unsigned char> empty; uint64_t emptyPixels = 0; float L[3] = { 0, 0, 0 }; uint32_t raysPerSample = 1; };
bool CullCells(const CullScene& DS, const DCamera& cam, int32_t maxPathLength, float rayTMin, uint32_t W, uint32_t H,
               uint32_t cellsX, uint32_t cellFirst, uint32_t stride, uint32_t numLocalCells, CullResult& out);
bool DeviceEvalMath(int fn, const float* x, const float* y, int n, float* out);
bool DeviceVerifyExactMath(int which, uint64_t* outMismatches, uint64_t* outFirst);   // 0: rtm::rcp1_ vs 1.0f / x, 1: rtm::sqrt_ vs sqrtf, 2: rtm::div_by_ vs a / b, 3: Barycentric short vs divisions; all 2^32 inputs
bool DeviceEvalHook(int kind, Scene* sc, const DCamera* cam, int a, int b, const float* in, int n, uint64_t seed, float* out);
void DeviceReleaseScene(DeviceScene* dev);

} // namespace rl
