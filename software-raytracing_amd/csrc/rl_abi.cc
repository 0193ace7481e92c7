// The C-ABI of the MI355X raylib: the 33 entry points of include/raylib.h (one per
// reference function, reference raylib/raylib.cc:25-331) plus the additional exports
// of include/raylib_amd.h.  Handles are raw pointers kept in mutex-guarded registries,
// as in the reference (raylib.cc:18-21, core/concurrent_vector.h:8-49).
#include "raylib.h"
#include "raylib_amd.h"
#include "rl_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>

using namespace rl;

namespace {

template <typename T>
struct Registry {
	std::mutex mu;
	std::vector<T*> items;
	void add(T* p) { std::lock_guard<std::mutex> lk(mu); items.push_back(p); }
	bool eraseFirst(T* p) {
		std::lock_guard<std::mutex> lk(mu);
		auto it = std::find(items.begin(), items.end(), p);
		if (it == items.end()) return false;
		items.erase(it);
		return true;
	}
	bool contains(T* p) { std::lock_guard<std::mutex> lk(mu); return std::find(items.begin(), items.end(), p) != items.end(); }
};
Registry<OBJModel> g_objModels;
Registry<Camera>   g_cameras;
Registry<Image>    g_images;
Registry<Scene>    g_scenes;
Registry<MaterialObj>  g_materials;
Registry<SceneElement> g_elements;

std::mutex g_stateMu;
uint64_t g_seed = 1;
bool g_seedSet = false;
RaylibAMDStats g_lastStats;

uint64_t CurrentSeed()
{
	std::lock_guard<std::mutex> lk(g_stateMu);
	if (!g_seedSet) {
		if (const char* e = getenv("RAYLIB_SEED")) g_seed = strtoull(e, nullptr, 10);
		g_seedSet = true;
	}
	return g_seed;
}

bool RenderInternal(const RendererSettings* settings, Scene* scene, Camera* camera,
                    uint32_t cellFirst, uint32_t cellStride, void* outDevice, float* outHost, bool callerOwnsOut = false)
{
	if (!settings || !scene || !camera) { Log("Raylib_Render: null argument"); return false; }
	if (!scene->finalized) { Log("Raylib_Render: scene was not finalized (Raylib_FinalizeScene)"); return false; }
	if (settings->renderMode >= RAYLIB_RENDERMODE_MAX) { Log("Raylib_Render: invalid render mode %u", settings->renderMode); return false; }
	if (scene->hasMovingCubes && (scene->accelT0 != camera->beginTime || scene->accelT1 != camera->endTime)) {
		// moving cubes: their boxes must cover the motion over THIS camera's shutter interval
		if (scene->device) { DeviceReleaseScene(scene->device); scene->device = nullptr; }
		if (!scene->BuildAccel(camera->beginTime, camera->endTime)) { Log("Raylib_Render: the acceleration structure could not be rebuilt for this camera's shutter interval"); return false; }
	}
	if (scene->sky && !g_images.contains(scene->sky)) {
		// the reference would read freed memory here; a destroyed panorama is treated as none
		Log("Raylib_Render: the scene's sky panorama was destroyed; rendering without it");
		scene->sky = nullptr;
	}
	RenderRequest req;
	req.settings = *settings;
	req.camera = camera->ToDevice();
	req.seed = CurrentSeed();
	req.cellFirst = cellFirst; req.cellStride = cellStride;
	req.outDevice = outDevice; req.outHostRGBA = outHost;
	req.callerOwnsOut = callerOwnsOut;
	RaylibAMDStats stats; memset(&stats, 0, sizeof(stats));
	bool ok = DeviceRender(*scene, req, stats);
	{ std::lock_guard<std::mutex> lk(g_stateMu); g_lastStats = stats; }
	return ok;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------
// reference raylib.cc:25-51
int32_t Raylib_Initialize(void)
{
	{ const char* q = getenv("RAYLIB_QUIET"); if (!(q && q[0] == '1')) printf("Initialize raylib\n"); }   // raylib.cc:27 prints this unconditionally
	LogStart();
	if (!DeviceAvailable()) {
		fprintf(stderr, "Raylib_Initialize: no usable HIP device (gfx950) -- this library has no CPU render path\n");
		return 0;
	}
	Log("Initialize obj loader");
	return 1;
}

int32_t Raylib_Terminate(void)
{
	{ const char* q = getenv("RAYLIB_QUIET"); if (!(q && q[0] == '1')) printf("Terminate raylib\n"); }
	Log("Destroy obj loader");
	LogStop();
	return 0;   // the reference returns 0 here despite its header comment (raylib.cc:50)
}

// ---------------------------------------------------------------------------
// reference raylib.cc:56-113
OBJModelHandle Raylib_LoadOBJModel(const char* objPath)
{
	OBJModel* m = new OBJModel;
	if (LoadOBJ(objPath, *m)) { g_objModels.add(m); return (OBJModelHandle)m; }
	delete m;
	return 0;
}

void Raylib_TransformOBJModel(OBJModelHandle h, float tx, float ty, float tz, float yaw, float pitch, float roll, float sx, float sy, float sz)
{
	if (!h) return;
	TransformOBJ(*(OBJModel*)h, tx, ty, tz, yaw, pitch, roll, sx, sy, sz);
}

void Raylib_FinalizeOBJModel(OBJModelHandle h)
{
	if (!h) return;
	((OBJModel*)h)->finalized = true;   // locks geometry (reference geom/static_mesh.cc:80-95); the BVH is built per scene
}

int32_t Raylib_UnloadOBJModel(OBJModelHandle h)
{
	OBJModel* m = (OBJModel*)h;
	if (g_objModels.eraseFirst(m)) { delete m; return 1; }
	return 0;
}

ImageHandle Raylib_LoadImage(const char* filepath)
{
	Image* img = LoadImageFile(filepath);
	if (!img) return 0;   // the reference registers a null image here (raylib.cc:108-113); returning NULL as its header documents
	g_images.add(img);
	return (ImageHandle)img;
}

// ---------------------------------------------------------------------------
// reference raylib.cc:118-179
CameraHandle Raylib_CreateCamera(void)
{
	Camera* c = new Camera;   // the reference leaves a default camera uninitialised (camera.h:14-21); this one is valid
	c->UpdateInternal();
	g_cameras.add(c);
	return (CameraHandle)c;
}
void Raylib_CameraSetPosition(CameraHandle h, float x, float y, float z) { Camera* c = (Camera*)h; if (!c) return; c->origin = F3(x, y, z); c->UpdateInternal(); }
void Raylib_CameraSetLookAt(CameraHandle h, float x, float y, float z) { Camera* c = (Camera*)h; if (!c) return; c->lookAt = F3(x, y, z); c->UpdateInternal(); }
void Raylib_CameraSetPerspective(CameraHandle h, float fovY, float aspect) { Camera* c = (Camera*)h; if (!c) return; c->fovY_degrees = fovY; c->aspectWH = aspect; c->UpdateInternal(); }
void Raylib_CameraSetLens(CameraHandle h, float aperture, float focal) { Camera* c = (Camera*)h; if (!c) return; c->aperture = aperture; c->focalDistance = focal; c->UpdateInternal(); }
void Raylib_CameraSetMotion(CameraHandle h, float t0, float t1) { Camera* c = (Camera*)h; if (!c) return; c->beginTime = t0; c->endTime = t1; c->UpdateInternal(); }
void Raylib_CameraCopy(CameraHandle src, CameraHandle dst) { if (!src || !dst) return; *(Camera*)dst = *(Camera*)src; }
int32_t Raylib_DestroyCamera(CameraHandle h)
{
	Camera* c = (Camera*)h;
	if (g_cameras.eraseFirst(c)) { delete c; return 1; }
	return 0;
}

// ---------------------------------------------------------------------------
// reference raylib.cc:181-203
ImageHandle Raylib_CreateImage(uint32_t width, uint32_t height)
{
	Image* img = new Image;
	img->Reallocate(width, height, 0.0f, 0.0f, 0.0f, 0.0f);   // Image2D(w, h, 0x0): ARGB 0 -> all channels 0
	g_images.add(img);
	return (ImageHandle)img;
}

void Raylib_DumpImageData(ImageHandle h, float* outDest)
{
	Image* img = (Image*)h;
	if (!img || !outDest) return;
	// a frame that lives on the device (Raylib_Render / Raylib_PostProcess left it there): packed to RGB there, 25 MB instead of 33 over the bus, through pinned staging
	if (img->hostStale && img->devValid && DeviceDumpRGB(*img, outDest)) return;
	img->SyncHost();   // the one read-back of a rendered / post-processed frame
	const size_t n = (size_t)img->width * img->height;
	for (size_t k = 0; k < n; ++k) {   // reference render/image.cc:121-135: packed RGB, row-major
		outDest[3 * k + 0] = img->rgba[4 * k + 0];
		outDest[3 * k + 1] = img->rgba[4 * k + 1];
		outDest[3 * k + 2] = img->rgba[4 * k + 2];
	}
}

int32_t Raylib_DestroyImage(ImageHandle h)
{
	Image* img = (Image*)h;
	if (g_images.eraseFirst(img)) { delete img; return 1; }
	return 0;
}

// ---------------------------------------------------------------------------
// reference raylib.cc:205-283
SceneHandle Raylib_CreateScene(void)
{
	Scene* s = new Scene;
	g_scenes.add(s);
	return (SceneHandle)s;
}

void Raylib_AddSceneElement(SceneHandle sh, SceneElementHandle eh)
{
	// The reference casts the handle to its C++ `Hitable*` (raylib.cc:258-262): a C++-ABI contract (vtables, class
	// layouts), not a C one.  Elements made by RaylibAMD_CreateSphere / Cube / Triangle are accepted; see INTEGRATION.md.
	Scene* s = (Scene*)sh; SceneElement* e = (SceneElement*)eh;
	if (!s || !e) return;
	if (!g_elements.contains(e)) {
		Log("Raylib_AddSceneElement: not an element created by this library (foreign C++ Hitable objects are not supported)");
		return;
	}
	if (!s->finalized) s->elements.push_back(e);   // reference geom/scene.cc:15-21: ignored after Finalize
}

void Raylib_AddOBJModelToScene(SceneHandle sh, OBJModelHandle oh)
{
	Scene* s = (Scene*)sh; OBJModel* m = (OBJModel*)oh;
	if (!s || !m) return;
	if (!s->finalized) s->models.push_back(m);   // reference geom/scene.cc:15-21: ignored after Finalize
}

void Raylib_SetSkyPanorama(SceneHandle sh, ImageHandle ih)
{
	Scene* s = (Scene*)sh;
	if (!s) return;
	// reference geom/scene.h keeps the handle and renderer.cc:159-176 reads the image through it at every miss: the panorama can be
	// set or replaced after Raylib_FinalizeScene and its pixels are those of render time.  Triangles and BVH are not touched.
	s->sky = (Image*)ih;
}
void Raylib_SetSunIlluminance(SceneHandle sh, float r, float g, float b)
{
	Scene* s = (Scene*)sh;
	if (!s) return;
	s->sunIlluminance = F3(r, g, b);
	if (s->device) { DeviceReleaseScene(s->device); s->device = nullptr; }
}
void Raylib_SetSunDirection(SceneHandle sh, float x, float y, float z)
{
	Scene* s = (Scene*)sh;
	if (!s) return;
	s->sunDirection = normalize(F3(x, y, z));   // reference geom/scene.h:20
	if (s->device) { DeviceReleaseScene(s->device); s->device = nullptr; }
}
void Raylib_FinalizeScene(SceneHandle sh)
{
	Scene* s = (Scene*)sh;
	if (!s) return;
	s->Finalize();
}
int32_t Raylib_DestroyScene(SceneHandle sh)
{
	Scene* s = (Scene*)sh;
	if (g_scenes.eraseFirst(s)) { delete s; return 1; }
	return 0;
}

// ---------------------------------------------------------------------------
// reference raylib.cc:231-239 -> render/renderer.cc:273-356
void Raylib_Render(const RendererSettings* settings, SceneHandle scene, CameraHandle camera, ImageHandle outMainImage)
{
	Image* img = (Image*)outMainImage;
	if (!settings || !img) { Log("Raylib_Render: null argument"); return; }
	if (settings->viewportWidth != img->width || settings->viewportHeight != img->height)
		img->Reallocate(settings->viewportWidth, settings->viewportHeight, 0.0f, 0.0f, 0.0f, 1.0f);   // renderer.cc:292-296
	if ((size_t)img->width * img->height == 0) return;
	// The frame stays on the device: Raylib_PostProcess works on it there, and the host pixels are fetched when somebody asks for them
	// (Raylib_DumpImageData, Raylib_WriteImageToDisk, ...).  A render that is REFUSED (scene not finalized, invalid mode, no device)
	// must leave the image as it was -- including a previous frame that still lives only on the device (hostStale): the image's state
	// changes only on success.  (Every refusal happens before anything is enqueued; DeviceImagePixels keeps a buffer that is large enough.)
	void* dev = DeviceImagePixels(*img);
	if (!RenderInternal(settings, (Scene*)scene, (Camera*)camera, 0, 1, dev, dev ? nullptr : img->rgba.data()))
		fprintf(stderr, "Raylib_Render: FAILED (no HIP device or invalid arguments); the image was not written\n");
	else { img->devValid = (dev != nullptr); img->hostStale = (dev != nullptr); img->Touch(); }
}

int32_t Raylib_Denoise(ImageHandle, int32_t, ImageHandle, ImageHandle, ImageHandle)
{
	return 0;   // reference render/renderer.cc:358-370 returns false when OIDN is not integrated (every non-Windows build)
}

void Raylib_PostProcess(ImageHandle h)
{
	Image* img = (Image*)h;
	if (!img) return;
	if (!DevicePostProcess(*img)) {
		Log("Raylib_PostProcess: no HIP device, running on the host");
		PostProcessHost(*img);
		img->Touch();
	}
}

int32_t Raylib_IsDenoiserSupported(void) { return 0; }

// ---------------------------------------------------------------------------
// reference raylib.cc:298-331
const char* Raylib_GetRenderModeString(uint32_t auxMode)
{
	static const char* names[] = { "Default", "Albedo", "SurfaceNormal", "MicrosurfaceNormal", "Texcoord", "Emission", "Reflectance" };
	return auxMode < RAYLIB_RENDERMODE_MAX ? names[auxMode] : nullptr;
}

int32_t Raylib_WriteImageToDisk(ImageHandle h, const char* filepath, uint32_t fileType)
{
	if (h == 0 || filepath == nullptr || fileType >= RAYLIB_IMAGEFILETYPE_MAX) return 0;
	return WriteImageFile(*(Image*)h, filepath, fileType) ? 1 : 0;
}

void Raylib_FlushLogThread(void) { LogFlush(); }

// ===========================================================================
// include/raylib_amd.h
// ===========================================================================
void RaylibAMD_SetSeed(uint64_t seed) { std::lock_guard<std::mutex> lk(g_stateMu); g_seed = seed; g_seedSet = true; }
uint64_t RaylibAMD_GetSeed(void) { return CurrentSeed(); }
void RaylibAMD_GetLastStats(RaylibAMDStats* out)
{
	if (!out) return;
	// a whole-frame render over several ranks returns with the frame in flight (rl_runtime.inl RenderMulti): its counters and times arrive now
	RaylibAMDStats late;
	const bool have = DeviceDrain(&late);
	std::lock_guard<std::mutex> lk(g_stateMu);
	if (have) g_lastStats = late;
	*out = g_lastStats;
}
int32_t RaylibAMD_DeviceAvailable(void) { return DeviceAvailable() ? 1 : 0; }
#ifndef RL_BUILD_ID
#define RL_BUILD_ID "unknown"
#endif
const char* RaylibAMD_BuildId(void) { return RL_BUILD_ID; }

uint32_t RaylibAMD_NumCells(uint32_t w, uint32_t h) { return ((w + 7) / 8) * ((h + 7) / 8); }
uint64_t RaylibAMD_CellBufferFloats(uint32_t w, uint32_t h, uint32_t cellFirst, uint32_t cellStride)
{
	if (cellStride <= 1 && cellFirst == 0) return (uint64_t)w * h * 4;
	const uint32_t n = RaylibAMD_NumCells(w, h);
	const uint32_t local = cellFirst < n ? (n - cellFirst + cellStride - 1) / cellStride : 0;
	return (uint64_t)local * 64 * 4;
}

int32_t RaylibAMD_RenderDevice(const RendererSettings* settings, SceneHandle scene, CameraHandle camera,
                               uint32_t cellFirst, uint32_t cellStride, void* outDevice)
{
	if (!settings || settings->viewportWidth == 0 || settings->viewportHeight == 0) return 0;
	// (a buffer of the caller's: nothing of the library's protects it while a frame is in flight, so this entry is synchronous on every path -- rl_runtime.inl RenderMulti)
	return RenderInternal(settings, (Scene*)scene, (Camera*)camera, cellFirst, cellStride ? cellStride : 1, outDevice, nullptr, outDevice != nullptr) ? 1 : 0;
}

int32_t RaylibAMD_RenderCellsHost(const RendererSettings* settings, SceneHandle scene, CameraHandle camera,
                                  uint32_t cellFirst, uint32_t cellStride, float* outHost)
{
	if (!settings || settings->viewportWidth == 0 || settings->viewportHeight == 0 || !outHost) return 0;
	return RenderInternal(settings, (Scene*)scene, (Camera*)camera, cellFirst, cellStride ? cellStride : 1, nullptr, outHost) ? 1 : 0;
}

MaterialHandle RaylibAMD_CreateMaterial(int32_t type, const float albedo[3], float roughness, float metallic,
                                        const float emissive[3], float ior, const float transmission[3], float fuzziness)
{
	if (type < 0 || type > MAT_DIFFUSE_LIGHT) return 0;
	MaterialObj* M = new MaterialObj; memset(&M->m, 0, sizeof(M->m));
	HostMaterial& m = M->m;
	m.type = type;
	for (int i = 0; i < 5; ++i) m.tex[i] = -1;
	auto clamp01 = [](float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); };
	for (int i = 0; i < 3; ++i) {
		m.albedo[i] = albedo ? albedo[i] : 0.0f;
		m.emissive[i] = emissive ? emissive[i] : 0.0f;
		m.transmission[i] = transmission ? transmission[i] : 1.0f;
	}
	m.roughness = roughness; m.metallic = metallic; m.ior = ior; m.fuzziness = fuzziness;
	// constructor-side clamps of the reference (render/material.h:79-82,107,236-238)
	if (type == MAT_LAMBERTIAN) for (int i = 0; i < 3; ++i) m.albedo[i] = clamp01(m.albedo[i]);
	if (type == MAT_METAL) m.fuzziness = clamp01(m.fuzziness);
	if (type == MAT_MICROFACET) { for (int i = 0; i < 3; ++i) m.albedo[i] = clamp01(m.albedo[i]); m.roughness = clamp01(m.roughness); m.metallic = clamp01(m.metallic); }
	g_materials.add(M);
	return (MaterialHandle)M;
}
int32_t RaylibAMD_DestroyMaterial(MaterialHandle h)
{
	MaterialObj* M = (MaterialObj*)h;
	if (g_materials.eraseFirst(M)) { delete M; return 1; }
	return 0;
}
static SceneElement* NewElement(PrimKind kind, MaterialHandle mh)
{
	MaterialObj* M = (MaterialObj*)mh;
	if (!M || !g_materials.contains(M)) return nullptr;
	SceneElement* e = new SceneElement; memset(e, 0, sizeof(*e));
	e->kind = kind; e->material = M;
	g_elements.add(e);
	return e;
}
SceneElementHandle RaylibAMD_CreateSphere(float cx, float cy, float cz, float radius, MaterialHandle material)
{
	SceneElement* e = NewElement(PRIM_SPHERE, material);
	if (!e) return 0;
	e->center = F3(cx, cy, cz); e->radius = radius;
	return (SceneElementHandle)e;
}
SceneElementHandle RaylibAMD_CreateCube(const float mn[3], const float mx[3], float timeStartMove, const float velocity[3], MaterialHandle material)
{
	if (!mn || !mx) return 0;
	SceneElement* e = NewElement(PRIM_CUBE, material);
	if (!e) return 0;
	e->minBounds = F3(mn[0], mn[1], mn[2]); e->maxBounds = F3(mx[0], mx[1], mx[2]); e->timeStartMove = timeStartMove;
	e->velocity = velocity ? F3(velocity[0], velocity[1], velocity[2]) : F3(0, 0, 0);
	return (SceneElementHandle)e;
}
SceneElementHandle RaylibAMD_CreateTriangle(const float v0[3], const float v1[3], const float v2[3],
                                            const float n0[3], const float n1[3], const float n2[3], const float uv[6], MaterialHandle material)
{
	if (!v0 || !v1 || !v2 || !n0 || !n1 || !n2) return 0;
	SceneElement* e = NewElement(PRIM_TRIANGLE, material);
	if (!e) return 0;
	HostTriangle& t = e->tri;
	t.v0 = F3(v0[0], v0[1], v0[2]); t.v1 = F3(v1[0], v1[1], v1[2]); t.v2 = F3(v2[0], v2[1], v2[2]);
	t.n0 = F3(n0[0], n0[1], n0[2]); t.n1 = F3(n1[0], n1[1], n1[2]); t.n2 = F3(n2[0], n2[1], n2[2]);
	if (uv) { t.s0 = uv[0]; t.t0 = uv[1]; t.s1 = uv[2]; t.t1 = uv[3]; t.s2 = uv[4]; t.t2 = uv[5]; }
	return (SceneElementHandle)e;
}
int32_t RaylibAMD_DestroySceneElement(SceneElementHandle h)
{
	SceneElement* e = (SceneElement*)h;
	if (g_elements.eraseFirst(e)) { delete e; return 1; }
	return 0;
}

int32_t RaylibAMD_EvalScatter(SceneHandle sh, int32_t material, const float* records, int32_t n, uint64_t seed, float* out)
{
	Scene* s = (Scene*)sh;
	if (!s || !s->finalized || !records || !out || material < 0 || material >= (int32_t)s->materials.size()) return 0;
	return DeviceEvalHook(0, s, nullptr, material, 0, records, n, seed, out) ? 1 : 0;
}
int32_t RaylibAMD_EvalCameraRays(CameraHandle ch, const float* uv, int32_t n, uint64_t seed, float* out)
{
	Camera* c = (Camera*)ch;
	if (!c || !uv || !out) return 0;
	const DCamera d = c->ToDevice();
	return DeviceEvalHook(1, nullptr, &d, 0, 0, uv, n, seed, out) ? 1 : 0;
}
int32_t RaylibAMD_CullCells(CameraHandle ch, const float* bounds, const float* sunIlluminance, const float* sunDirection, int32_t width, int32_t height,
                            uint8_t* outEmpty, float* outConstant)
{
	Camera* c = (Camera*)ch;
	if (!c || !bounds || width <= 0 || height <= 0) return -1;
	CullScene cs;
	for (int k = 0; k < 3; ++k) { cs.boundsMin[k] = bounds[k]; cs.boundsMax[k] = bounds[3 + k]; }
	cs.boundsValid = true;
	if (sunIlluminance && sunDirection) {
		for (int k = 0; k < 3; ++k) { cs.sunIlluminance[k] = sunIlluminance[k]; cs.sunDirection[k] = sunDirection[k]; }
		cs.hasSun = !(sunIlluminance[0] == 0.0f && sunIlluminance[1] == 0.0f && sunIlluminance[2] == 0.0f);
	}
	const uint32_t W = (uint32_t)width, H = (uint32_t)height, cellsX = (W + 7) / 8, cellsY = (H + 7) / 8;
	CullResult r;
	if (!CullCells(cs, c->ToDevice(), 1, 0.0f, W, H, cellsX, 0, 1, cellsX * cellsY, r)) {
		if (outEmpty) memset(outEmpty, 0, (size_t)cellsX * cellsY);
		return r.empty.empty() ? -1 : 0;   // not eligible, or eligible with nothing to drop
	}
	if (outEmpty) memcpy(outEmpty, r.empty.data(), (size_t)cellsX * cellsY);
	if (outConstant) { outConstant[0] = r.L[0]; outConstant[1] = r.L[1]; outConstant[2] = r.L[2]; }
	return (int32_t)(cellsX * cellsY - (uint32_t)r.active.size());
}
int32_t RaylibAMD_EvalTexture(SceneHandle sh, int32_t texture, int32_t bSRGB, const float* uv, int32_t n, float* out)
{
	Scene* s = (Scene*)sh;
	if (!s || !s->finalized || !uv || !out || texture < 0 || texture >= (int32_t)s->textures.size()) return 0;
	return DeviceEvalHook(2, s, nullptr, texture, bSRGB, uv, n, 0, out) ? 1 : 0;
}

int32_t RaylibAMD_EvalDeviceMath(int32_t fn, const float* x, const float* y, int32_t n, float* out)
{
	if (!x || !out) return 0;
	return DeviceEvalMath(fn, x, y, n, out) ? 1 : 0;
}

int32_t RaylibAMD_VerifyExactMath(int32_t which, uint64_t* outMismatches, uint64_t* outFirstBits)
{
	if (which < 0 || which > 3) return 0;
	return DeviceVerifyExactMath(which, outMismatches, outFirstBits) ? 1 : 0;
}

int32_t RaylibAMD_ClosestHit(SceneHandle sh, const float* rays, int32_t n, float tMin, void* outHits)
{
	Scene* s = (Scene*)sh;
	if (!s || !s->finalized || !rays || !outHits) return 0;
	return DeviceClosestHit(*s, rays, n, tMin, outHits) ? 1 : 0;
}

int32_t RaylibAMD_SceneNumTriangles(SceneHandle sh) { Scene* s = (Scene*)sh; return s ? (int32_t)s->triangles.size() : 0; }
int32_t RaylibAMD_SceneNumMaterials(SceneHandle sh) { Scene* s = (Scene*)sh; return s ? (int32_t)s->materials.size() : 0; }
int32_t RaylibAMD_SceneNumTextures(SceneHandle sh) { Scene* s = (Scene*)sh; return s ? (int32_t)s->textures.size() : 0; }
void RaylibAMD_SceneExportTriangles(SceneHandle sh, void* out) { Scene* s = (Scene*)sh; if (s && out && !s->triangles.empty()) memcpy(out, s->triangles.data(), s->triangles.size() * sizeof(HostTriangle)); }
void RaylibAMD_SceneExportMaterials(SceneHandle sh, void* out) { Scene* s = (Scene*)sh; if (s && out && !s->materials.empty()) memcpy(out, s->materials.data(), s->materials.size() * sizeof(HostMaterial)); }
void RaylibAMD_SceneTextureSize(SceneHandle sh, int32_t i, int32_t* w, int32_t* h)
{
	Scene* s = (Scene*)sh;
	if (!s || i < 0 || i >= (int32_t)s->textures.size()) { if (w) *w = 0; if (h) *h = 0; return; }
	if (w) *w = (int32_t)s->textures[i]->width;
	if (h) *h = (int32_t)s->textures[i]->height;
}
void RaylibAMD_SceneExportTexture(SceneHandle sh, int32_t i, float* out)
{
	Scene* s = (Scene*)sh;
	if (!s || !out || i < 0 || i >= (int32_t)s->textures.size()) return;
	s->textures[i]->SyncHost();
	memcpy(out, s->textures[i]->rgba.data(), s->textures[i]->rgba.size() * sizeof(float));
}
void RaylibAMD_SceneGetSun(SceneHandle sh, float ill[3], float dir[3])
{
	Scene* s = (Scene*)sh;
	if (!s) return;
	ill[0] = s->sunIlluminance.x; ill[1] = s->sunIlluminance.y; ill[2] = s->sunIlluminance.z;
	dir[0] = s->sunDirection.x; dir[1] = s->sunDirection.y; dir[2] = s->sunDirection.z;
}
int32_t RaylibAMD_SceneBVHInfo(SceneHandle sh, uint32_t* nodes, uint32_t* depth, float* sah)
{
	Scene* s = (Scene*)sh;
	if (!s || !s->finalized) return 0;
	if (nodes) *nodes = (uint32_t)s->bvh.nodes.size();
	if (depth) *depth = s->bvh.depth;
	if (sah) *sah = s->bvh.sahCost;
	return ValidateBVH(s->bvh, s->triangles) ? 1 : 0;
}
int32_t RaylibAMD_SceneBVH4Info(SceneHandle sh, uint32_t* nodes4, uint32_t* stackNeed)
{
	Scene* s = (Scene*)sh;
	if (!s || !s->finalized || s->bvh.nodes4.empty()) return 0;
	if (nodes4) *nodes4 = (uint32_t)s->bvh.nodes4.size();
	if (stackNeed) *stackNeed = s->bvh.stackNeed4;
	return (ValidateBVH4(s->bvh, s->triangles) && ValidateBVH8(s->bvh, s->triangles)) ? 1 : -1;   // (the 8-wide tree, where the scene has one, is part of the check)
}
int32_t RaylibAMD_SceneBVH8Info(SceneHandle sh, uint32_t* nodes8, uint32_t* levels, float* steps4, float* steps8)
{
	Scene* s = (Scene*)sh;
	if (!s || !s->finalized || s->bvh.nodes8.empty()) return 0;
	if (nodes8) *nodes8 = (uint32_t)s->bvh.nodes8.size();
	if (levels) *levels = s->bvh.depth8;
	if (steps4) *steps4 = s->bvh.sahNodes4;
	if (steps8) *steps8 = s->bvh.sahNodes8;
	return ValidateBVH8(s->bvh, s->triangles) ? 1 : -1;
}
int32_t RaylibAMD_SceneWalk8Host(SceneHandle sh, const float* rays, int32_t count, float tMin, const float* tMax, float* outT, uint32_t* outSteps)
{
	Scene* s = (Scene*)sh;
	if (!s || !s->finalized || s->bvh.nodes8.empty() || !rays || !tMax || !outT || count < 0) return 0;
	return Walk8Host(s->bvh, s->triangles, rays, count, tMin, tMax, outT, outSteps) ? 1 : -1;
}
int32_t RaylibAMD_SceneLeafListInfo(SceneHandle sh, uint32_t* maxPerLeaf)
{
	Scene* s = (Scene*)sh;
	if (!s || !s->finalized) return 0;
	int32_t leaves = 0; uint32_t most = 0;
	for (const DNode4& n : s->bvh.leafList) for (int k = 0; k < 4; ++k) if (n.child[k] != DNODE_EMPTY) { ++leaves; most = std::max(most, (((uint32_t)~n.child[k]) & 7u) + 1u); }
	if (maxPerLeaf) *maxPerLeaf = most;
	return leaves;
}
uint64_t RaylibAMD_SceneBVHHash(SceneHandle sh)
{
	Scene* s = (Scene*)sh;
	if (!s || !s->finalized) return 0;
	uint64_t h = 1469598103934665603ull;   // FNV-1a over the node records and the leaf order
	auto feed = [&h](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
	feed(s->bvh.nodes.data(), s->bvh.nodes.size() * sizeof(DNode));
	feed(s->bvh.triOrder.data(), s->bvh.triOrder.size() * sizeof(uint32_t));
	feed(s->bvh.nodes4.data(), s->bvh.nodes4.size() * sizeof(DNode4));
	feed(s->bvh.nodes4q.data(), s->bvh.nodes4q.size() * sizeof(DNode4Q));
	feed(s->bvh.leafList.data(), s->bvh.leafList.size() * sizeof(DNode4));
	feed(s->bvh.nodes8.data(), s->bvh.nodes8.size() * sizeof(DNode8));
	return h;
}
void RaylibAMD_CameraExport(CameraHandle h, float out[19])
{
	Camera* c = (Camera*)h;
	if (!c || !out) return;
	const f3* v[6] = { &c->origin, &c->top_left, &c->horizontal, &c->vertical, &c->u, &c->v };
	int k = 0;
	out[k++] = c->origin.x; out[k++] = c->origin.y; out[k++] = c->origin.z; out[k++] = c->lensRadius;
	for (int i = 1; i < 6; ++i) { out[k++] = v[i]->x; out[k++] = v[i]->y; out[k++] = v[i]->z; }
}
ImageHandle RaylibAMD_CreateImageFromData(uint32_t w, uint32_t h, const float* rgba)
{
	if (!rgba) return 0;
	Image* img = new Image;
	img->width = w; img->height = h;
	img->rgba.assign(rgba, rgba + (size_t)w * h * 4);
	g_images.add(img);
	return (ImageHandle)img;
}
void RaylibAMD_DumpImageRGBA(ImageHandle h, float* out)
{
	Image* img = (Image*)h;
	if (!img || !out) return;
	img->SyncHost();
	memcpy(out, img->rgba.data(), img->rgba.size() * sizeof(float));
}
float RaylibAMD_ParseFloat(const char* token) { return token ? ParseDecimalFloat(token) : 0.0f; }
int32_t RaylibAMD_ImageSize(ImageHandle h, uint32_t* w, uint32_t* ht)
{
	Image* img = (Image*)h;
	if (!img) return 0;
	if (w) *w = img->width;
	if (ht) *ht = img->height;
	return 1;
}
int32_t RaylibAMD_OBJModelSetTexture(OBJModelHandle oh, const char* materialName, int32_t slot, ImageHandle ih)
{
	OBJModel* m = (OBJModel*)oh; Image* img = (Image*)ih;
	if (!m || !materialName || slot < 0 || slot > 4 || !img || m->finalized) return 0;
	for (size_t i = 0; i < m->materialNames.size(); ++i) {
		if (m->materialNames[i] == materialName && m->materials[i].type == MAT_MICROFACET) {
			m->images.push_back(std::make_shared<Image>(*img));
			m->materials[i].tex[slot] = (int32_t)m->images.size() - 1;
			return 1;
		}
	}
	return 0;
}

} // extern "C"
