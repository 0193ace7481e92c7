// TEST INFRASTRUCTURE -- driver that exposes the REAL reference renderer
// (compiled in place from /root/reference/raylib, never copied) through a small
// C API so tests/ and bench.py's cpu_baseline leg can call it with ctypes.
//
// What in oracle/_ref/*.so is reference code and what is not:
//   reference, unmodified, compiled from /root/reference/raylib:
//     core/{random,thread_pool,logger,assertion}.cc
//     geom/{triangle,bvh,hit,scene,sphere,cube,transform,static_mesh}.cc
//     render/{material,texture,renderer}.cc
//   NOT reference:
//     - this file (scene assembly that mirrors loader/obj_loader.cc:133-245 and
//       raylib.cc:205-283; the per-sample loop that mirrors renderer.cc:229-248)
//     - oracle/ref_shim/core/random.h (seeded build only; see that file)
//     - the six Image2D container methods below: render/image.cc cannot be built
//       here (it includes FreeImage.h, a third-party header this image lacks,
//       and no stand-in for it is written), so the pixel container's
//       ctor/Reallocate/SetPixel are defined here.  They hold no arithmetic.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this.

#include "flat_scene.h"

#include "core/random.h"
#include "core/vec3.h"
#include "geom/ray.h"
#include "geom/hit.h"
#include "geom/bvh.h"
#include "geom/triangle.h"
#include "geom/static_mesh.h"
#include "geom/scene.h"
#include "geom/sphere.h"
#include "geom/cube.h"
#include "render/camera.h"
#include "render/material.h"
#include "render/texture.h"
#include "render/image.h"
#include "render/renderer.h"

#include <map>
#include <memory>
#include <thread>
#include <vector>
#include <atomic>
#include <cstring>

// ---------------------------------------------------------------------------
// RNG stream state for the seeded build (declared in ref_shim/core/random.h).
#ifndef REF_NATIVE_RNG
thread_local RaylibRngStream g_refRngStream = {0};
thread_local uint64_t g_refRngDraws = 0;
#define REF_RNG_SELECT(seed, px, s) RefRngSelect((seed), (px), (s))
#else
#define REF_RNG_SELECT(seed, px, s) ((void)0)
#endif

// ---------------------------------------------------------------------------
// Image2D container methods (see header comment; declarations are the
// reference's own, render/image.h:88-119).
Image2D::Image2D() { Reallocate(0, 0); }
Image2D::Image2D(uint32 w, uint32 h, const Pixel& px) { Reallocate(w, h, px); }
Image2D::Image2D(uint32 w, uint32 h, uint32 color) : Image2D(w, h, Pixel(color)) {}
void Image2D::Reallocate(uint32 w, uint32 h, const Pixel& clearColor)
{
	width = w;
	height = h;
	image.resize((size_t)w * h, clearColor);
}
void Image2D::SetPixel(int32 x, int32 y, const Pixel& px) { image[ix(x, y)] = px; }
void Image2D::SetPixel(int32 x, int32 y, uint32 argb) { image[ix(x, y)] = Pixel(argb); }

// ---------------------------------------------------------------------------
// Link fix: reference core/logger.cc:30 declares `void LogMain();` at block scope
// inside namespace Logger (so g++ looks for Logger::LogMain) but defines it at
// global scope (:74).  Forward one to the other; the log thread is never started here.
void LogMain();
namespace Logger { void LogMain() { ::LogMain(); } }

// ---------------------------------------------------------------------------
// Symbols with external linkage that live in the reference's renderer.cc.
struct RayPayload { // layout of reference render/renderer.cc:48-51
	int32 maxRecursion;
	float rayTMin;
};
vec3 TraceScene(const ray& pathRay, const Scene* world, int depth, const RayPayload& settings); // renderer.cc:114
vec3 TraceSceneDebugMode(const ray& pathRay, const Scene* world, const RayPayload& settings, ERenderMode debugMode); // renderer.cc:62

namespace {

inline vec3 V(const float* f) { return vec3(f[0], f[1], f[2]); }

struct RefScene {
	Scene scene;
	std::vector<Material*> materials;
	std::map<const Material*, int32_t> materialIndex;
	std::vector<std::shared_ptr<Image2D>> images;
	std::vector<StaticMesh*> meshes;
	Hitable* root = nullptr;
	BVHNode* shapeBVH = nullptr;
};

Camera MakeCamera(const FlatCamera* c)
{
	return Camera(V(c->origin), V(c->lookAt), c->fovY_degrees, c->aspectWH,
	              c->aperture, c->focalDistance, c->beginTime, c->endTime);
}

void FillHit(const RefScene* rs, bool bHit, const HitResult& h, FlatHit* out)
{
	memset(out, 0, sizeof(*out));
	out->hit = bHit ? 1 : 0;
	out->material = -1;
	if (!bHit) return;
	out->t = h.t;
	out->p[0] = h.p.x; out->p[1] = h.p.y; out->p[2] = h.p.z;
	out->n[0] = h.n.x; out->n[1] = h.n.y; out->n[2] = h.n.z;
	out->paramU = h.paramU;
	out->paramV = h.paramV;
	auto it = rs->materialIndex.find(h.material);
	out->material = (it == rs->materialIndex.end()) ? -1 : it->second;
}

} // namespace

extern "C" {

// Build the object graph exactly as the reference's OBJ path does:
// one StaticMesh per shape (obj_loader.cc:133-234), root = the mesh or a
// shape-level BVHNode built BEFORE the meshes are finalized (obj_loader.cc:236-245),
// per-mesh Finalize (raylib.cc:92-95), then Scene::Finalize (raylib.cc:212-215).
void* ref_scene_create(const FlatSceneDesc* desc, uint64_t buildSeed)
{
	REF_RNG_SELECT(buildSeed, RAYLIB_RNG_BUILD_PIXEL, 0);

	RefScene* rs = new RefScene;

	for (int32_t i = 0; i < desc->numTextures; ++i) {
		const FlatTexture& t = desc->textures[i];
		auto img = std::make_shared<Image2D>((uint32)t.width, (uint32)t.height, Pixel(0.0f, 0.0f, 0.0f, 0.0f));
		for (int32_t y = 0; y < t.height; ++y)
			for (int32_t x = 0; x < t.width; ++x) {
				const float* p = t.rgba + 4 * ((size_t)y * t.width + x);
				img->SetPixel(x, y, Pixel(p[0], p[1], p[2], p[3]));
			}
		rs->images.push_back(img);
	}
	auto Tex = [&](int32_t ix) { return (ix >= 0 && ix < (int32_t)rs->images.size()) ? rs->images[ix] : std::shared_ptr<Image2D>(); };

	for (int32_t i = 0; i < desc->numMaterials; ++i) {
		const FlatMaterial& m = desc->materials[i];
		Material* M = nullptr;
		switch (m.type) {
			case FLAT_MAT_LAMBERTIAN:    M = new Lambertian(V(m.albedo)); break;
			case FLAT_MAT_MIRROR:        M = new Mirror(V(m.albedo)); break;
			case FLAT_MAT_DIELECTRIC:    M = new Dielectric(m.ior, V(m.transmission)); break;
			case FLAT_MAT_METAL:         M = new Metal(V(m.albedo), m.fuzziness); break;
			case FLAT_MAT_DIFFUSE_LIGHT: M = new DiffuseLight(V(m.albedo)); break;
			default: {
				// same setter sequence as obj_loader.cc:372-395
				MicrofacetMaterial* mm = new MicrofacetMaterial;
				if (Tex(m.texAlbedo))    mm->SetAlbedoTexture(Tex(m.texAlbedo));
				if (Tex(m.texNormal))    mm->SetNormalTexture(Tex(m.texNormal));
				if (Tex(m.texRoughness)) mm->SetRoughnessTexture(Tex(m.texRoughness));
				if (Tex(m.texMetallic))  mm->SetMetallicTexture(Tex(m.texMetallic));
				if (Tex(m.texEmissive))  mm->SetEmissiveTexture(Tex(m.texEmissive));
				mm->SetAlbedoFallback(V(m.albedo));
				mm->SetRoughnessFallback(m.roughness);
				mm->SetMetallicFallback(m.metallic);
				mm->SetEmissiveFallback(V(m.emissive));
				M = mm;
			}
		}
		rs->materials.push_back(M);
		rs->materialIndex[M] = i;
	}

	int32_t nShapes = desc->numShapes > 0 ? desc->numShapes : 1;
	if (desc->numTriangles == 0) nShapes = 0;
	rs->meshes.resize(nShapes);
	for (int32_t s = 0; s < nShapes; ++s) rs->meshes[s] = new StaticMesh;
	for (int32_t i = 0; i < desc->numTriangles; ++i) {
		const FlatTriangle& f = desc->triangles[i];
		Triangle T(V(f.v0), V(f.v1), V(f.v2), V(f.n0), V(f.n1), V(f.n2), rs->materials[f.material]);
		T.SetParameterization(f.s0, f.t0, f.s1, f.t1, f.s2, f.t2);
		rs->meshes[f.shape]->AddTriangle(T);
	}
	for (StaticMesh* m : rs->meshes) m->CalculateBounds();

	if (nShapes == 0) {
		rs->root = nullptr;
	} else if (nShapes == 1) {
		rs->root = rs->meshes[0];
	} else {
		std::vector<Hitable*> hs(rs->meshes.begin(), rs->meshes.end());
		rs->shapeBVH = new BVHNode(new HitableList(hs), 0.0f, 0.0f);
		rs->root = rs->shapeBVH;
	}
	for (StaticMesh* m : rs->meshes) m->Finalize();

	if (desc->numTriangles > 0) rs->scene.AddSceneElement(rs->root);
	for (int32_t i = 0; i < desc->numSpheres; ++i) {
		const FlatSphere& f = desc->spheres[i];
		rs->scene.AddSceneElement(new Sphere(V(f.center), f.radius, rs->materials[f.material]));
	}
	for (int32_t i = 0; i < desc->numCubes; ++i) {
		const FlatCube& f = desc->cubes[i];
		rs->scene.AddSceneElement(new Cube(V(f.minBounds), V(f.maxBounds), f.timeStartMove, V(f.velocity), rs->materials[f.material]));
	}
	rs->scene.SetSunIlluminance(V(desc->sunIlluminance));
	rs->scene.SetSunDirection(V(desc->sunDirection));
	if (desc->skyTexture >= 0) rs->scene.SetSkyPanorama((ImageHandle)rs->images[desc->skyTexture].get());
	rs->scene.Finalize();
	return rs;
}

void ref_scene_destroy(void* h)
{
	// The reference never frees BVH interiors / materials either; tests are short-lived.
	(void)h;
}

// Seeded render.  Per pixel and sample this re-keys the stream and then runs
// the body of GenerateCell (reference renderer.cc:229-248 default mode,
// :258-268 debug modes) around the reference's Camera::GetCameraRay and TraceScene.
// outRGBA: H*W*4 floats (alpha 1, as Pixel(r,g,b) does). outSamples (optional): H*W*SPP*3.
// Region form: only pixels [x0, x0+rw) x [y0, y0+rh) of the W x H image are computed;
// outRGBA is rw*rh*4 floats, outSamples (optional) rw*rh*SPP*3.  Pixel keys stay those
// of the full image, so a window of a large render can be checked without rendering it all.
void ref_render_region(void* h, const FlatCamera* fc, const FlatSettings* st, uint64_t seed,
                       int32_t numThreads, int32_t x0, int32_t y0, int32_t rw, int32_t rh,
                       float* outRGBA, float* outSamples)
{
	RefScene* rs = (RefScene*)h;
	const Camera camera = MakeCamera(fc);
	const int32_t W = (int32_t)st->viewportWidth, H = (int32_t)st->viewportHeight;
	const float imageWidth = (float)W, imageHeight = (float)H;
	const int32_t SPP = std::max(1, st->samplesPerPixel);
	const RayPayload rt{ st->maxPathLength, st->rayTMin };
	const bool bDefault = (st->renderMode == RAYLIB_RENDERMODE_Default);
	if (numThreads < 1) numThreads = 1;

	std::atomic<int32_t> nextRow(y0);
	auto worker = [&]() {
		for (;;) {
			int32_t y = nextRow.fetch_add(1);
			if (y >= y0 + rh || y >= H) break;
			for (int32_t x = x0; x < x0 + rw && x < W; ++x) {
				const uint32_t pixelIndex = (uint32_t)(y * W + x);
				const size_t outIndex = (size_t)(y - y0) * rw + (x - x0);
				vec3 result;
				if (bDefault) {
					static thread_local RNG randomsAA(4096 * 8);
					vec3 accum(0.0f, 0.0f, 0.0f);
					for (int32_t s = 0; s < SPP; ++s) {
						REF_RNG_SELECT(seed, pixelIndex, (uint32_t)s);
						float u = (float)x / imageWidth;
						float v = (float)y / imageHeight;
						if (s != 0) {
							u += (randomsAA.Peek() - 0.5f) * 2.0f / imageWidth;
							v += (randomsAA.Peek() - 0.5f) * 2.0f / imageHeight;
						}
						ray cameraRay = camera.GetCameraRay(u, v);
						vec3 Li = TraceScene(cameraRay, &rs->scene, 0, rt);
						accum += Li;
						if (outSamples) {
							float* o = outSamples + 3 * (outIndex * SPP + s);
							o[0] = Li.x; o[1] = Li.y; o[2] = Li.z;
						}
					}
					accum /= (float)SPP;
					result = accum;
				} else {
					REF_RNG_SELECT(seed, pixelIndex, 0);
					float u = (float)x / imageWidth;
					float v = (float)y / imageHeight;
					ray cameraRay = camera.GetCameraRay(u, v);
					result = TraceSceneDebugMode(cameraRay, &rs->scene, rt, (ERenderMode)st->renderMode);
				}
				float* o = outRGBA + 4 * outIndex;
				o[0] = result.x; o[1] = result.y; o[2] = result.z; o[3] = 1.0f;
			}
		}
	};
	std::vector<std::thread> threads;
	for (int32_t i = 1; i < numThreads; ++i) threads.emplace_back(worker);
	worker();
	for (auto& t : threads) t.join();
}

void ref_render(void* h, const FlatCamera* fc, const FlatSettings* st, uint64_t seed,
                int32_t numThreads, float* outRGBA, float* outSamples)
{
	ref_render_region(h, fc, st, seed, numThreads, 0, 0, (int32_t)st->viewportWidth, (int32_t)st->viewportHeight, outRGBA, outSamples);
}

// The reference's own entry point, untouched: Renderer::RenderScene with its
// ThreadPool of hardware_concurrency() threads (reference renderer.cc:273-356).
// Used for cpu_baseline timing (kind "reference") from the native-RNG build.
void ref_render_native(void* h, const FlatCamera* fc, const FlatSettings* st, float* outRGBA)
{
	RefScene* rs = (RefScene*)h;
	Camera camera = MakeCamera(fc);
	RendererSettings settings;
	settings.viewportWidth = st->viewportWidth;
	settings.viewportHeight = st->viewportHeight;
	settings.samplesPerPixel = st->samplesPerPixel;
	settings.maxPathLength = st->maxPathLength;
	settings.rayTMin = st->rayTMin;
	settings.renderMode = st->renderMode;
	Image2D image(st->viewportWidth, st->viewportHeight, 0x0);
	Renderer renderer;
	renderer.RenderScene(&settings, &rs->scene, &camera, &image);
	if (outRGBA) {
		const std::vector<Pixel>& px = image.GetPixelArray();
		memcpy(outRGBA, px.data(), px.size() * sizeof(Pixel));
	}
}

// ---- known-answer helpers --------------------------------------------------

// rays: n * 6 floats (origin, direction). Closest hit through the whole
// nested accel structure (reference renderer.cc:129).
void ref_closest_hit(void* h, const float* rays, int32_t n, float tMin, FlatHit* out)
{
	RefScene* rs = (RefScene*)h;
	for (int32_t i = 0; i < n; ++i) {
		ray r(V(rays + 6 * i), V(rays + 6 * i + 3), 0.0f);
		HitResult hit;
		bool b = rs->scene.GetAccelStruct()->Hit(r, tMin, FLOAT_MAX, hit);
		FillHit(rs, b, hit, out + i);
	}
}

// boxes: n * 6 floats (min, max); rays: n * 6 floats. reference geom/aabb.h:14-55
void ref_aabb_hit(const float* boxes, const float* rays, int32_t n, float tMin, float tMax, int32_t* out)
{
	for (int32_t i = 0; i < n; ++i) {
		AABB box(V(boxes + 6 * i), V(boxes + 6 * i + 3));
		ray r(V(rays + 6 * i), V(rays + 6 * i + 3), 0.0f);
		out[i] = box.Hit(r, tMin, tMax) ? 1 : 0;
	}
}

// Single triangle test (reference geom/triangle.cc:18-58) against a Lambertian.
void ref_triangle_hit(const FlatTriangle* tris, const float* rays, int32_t n, float tMin, float tMax, FlatHit* out)
{
	static Lambertian dummy(vec3(0.5f));
	for (int32_t i = 0; i < n; ++i) {
		const FlatTriangle& f = tris[i];
		Triangle T(V(f.v0), V(f.v1), V(f.v2), V(f.n0), V(f.n1), V(f.n2), &dummy);
		T.SetParameterization(f.s0, f.t0, f.s1, f.t1, f.s2, f.t2);
		ray r(V(rays + 6 * i), V(rays + 6 * i + 3), 0.0f);
		HitResult hit;
		bool b = T.Hit(r, tMin, tMax, hit);
		memset(out + i, 0, sizeof(FlatHit));
		out[i].hit = b;
		out[i].material = -1;
		if (b) {
			out[i].t = hit.t;
			out[i].p[0] = hit.p.x; out[i].p[1] = hit.p.y; out[i].p[2] = hit.p.z;
			out[i].n[0] = hit.n.x; out[i].n[1] = hit.n.y; out[i].n[2] = hit.n.z;
			out[i].paramU = hit.paramU; out[i].paramV = hit.paramV;
		}
	}
}

// normals: n*3; vecs: n*3 -> outLocal n*3 (WorldToLocal), outWorld n*3 (LocalToWorld). reference geom/hit.cc:6-30
void ref_onb(const float* normals, const float* vecs, int32_t n, float* outLocal, float* outWorld)
{
	for (int32_t i = 0; i < n; ++i) {
		HitResult h;
		h.n = V(normals + 3 * i);
		h.BuildOrthonormalBasis();
		vec3 l = h.WorldToLocal(V(vecs + 3 * i));
		vec3 w = h.LocalToWorld(V(vecs + 3 * i));
		outLocal[3 * i] = l.x; outLocal[3 * i + 1] = l.y; outLocal[3 * i + 2] = l.z;
		outWorld[3 * i] = w.x; outWorld[3 * i + 1] = w.y; outWorld[3 * i + 2] = w.z;
	}
}

// Camera rays: uv n*2 -> out n*7 (o, d, time); stream (seed, i, 0) per ray. reference render/camera.h:44-53
void ref_camera_rays(const FlatCamera* fc, const float* uv, int32_t n, uint64_t seed, float* out)
{
	Camera camera = MakeCamera(fc);
	for (int32_t i = 0; i < n; ++i) {
		REF_RNG_SELECT(seed, (uint32_t)i, 0);
		ray r = camera.GetCameraRay(uv[2 * i], uv[2 * i + 1]);
		float* o = out + 7 * i;
		o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z; o[6] = r.t;
	}
}

// One Scatter/ScatteringPdf/Emitted evaluation per record, stream (seed, i, 0).
// in : n * 16 floats = ray o(3) d(3) time(1), hit t(1) p(3) n(3) paramU paramV
// out: n * 16 floats = scattered(1) refl(3) dir(3) origin(3) pdf(1) scatteringPdf(1) emitted(3) draws(1)
void ref_scatter(void* h, int32_t material, const float* in, int32_t n, uint64_t seed, float* out)
{
	RefScene* rs = (RefScene*)h;
	Material* M = rs->materials[material];
	for (int32_t i = 0; i < n; ++i) {
		const float* a = in + 16 * i;
		ray r(V(a), V(a + 3), a[6]);
		HitResult hit;
		hit.t = a[7];
		hit.p = V(a + 8);
		hit.n = V(a + 11);
		hit.paramU = a[14];
		hit.paramV = a[15];
		hit.material = M;
		hit.BuildOrthonormalBasis();
		REF_RNG_SELECT(seed, (uint32_t)i, 0);
		vec3 refl(0.0f);
		ray sc;
		float pdf = 0.0f;
		bool b = M->Scatter(r, hit, refl, sc, pdf);
		float sp = b ? M->ScatteringPdf(hit, -r.d, sc.d) : 0.0f;
		vec3 e = M->Emitted(hit, r.d);
		float* o = out + 16 * i;
		o[0] = b ? 1.0f : 0.0f;
		o[1] = refl.x; o[2] = refl.y; o[3] = refl.z;
		o[4] = sc.d.x; o[5] = sc.d.y; o[6] = sc.d.z;
		o[7] = sc.o.x; o[8] = sc.o.y; o[9] = sc.o.z;
		o[10] = pdf; o[11] = sp;
		o[12] = e.x; o[13] = e.y; o[14] = e.z;
#ifndef REF_NATIVE_RNG
		o[15] = (float)g_refRngDraws;
#else
		o[15] = -1.0f;
#endif
	}
}

// Texture2D::Sample (reference render/texture.cc:30-53). uv n*2 -> out n*4.
void ref_texture_sample(const FlatTexture* t, int32_t bSRGB, const float* uv, int32_t n, float* out)
{
	auto img = std::make_shared<Image2D>((uint32)t->width, (uint32)t->height, Pixel(0.0f, 0.0f, 0.0f, 0.0f));
	for (int32_t y = 0; y < t->height; ++y)
		for (int32_t x = 0; x < t->width; ++x) {
			const float* p = t->rgba + 4 * ((size_t)y * t->width + x);
			img->SetPixel(x, y, Pixel(p[0], p[1], p[2], p[3]));
		}
	Texture2D* tex = Texture2D::CreateFromImage2D(img);
	SamplerState ss;
	ss.bSRGB = (bSRGB != 0);
	tex->SetSamplerState(ss);
	for (int32_t i = 0; i < n; ++i) {
		Pixel p = tex->Sample(uv[2 * i], uv[2 * i + 1]);
		out[4 * i] = p.r; out[4 * i + 1] = p.g; out[4 * i + 2] = p.b; out[4 * i + 3] = p.a;
	}
	delete tex;
}

// Walk the reference's object graph and report its shape (nodes, depth).
static void WalkBVH(const Hitable* h, int depth, int64_t* nodes, int32_t* maxDepth)
{
	if (const BVHNode* b = dynamic_cast<const BVHNode*>(h)) {
		*nodes += 1;
		if (depth > *maxDepth) *maxDepth = depth;
		WalkBVH(b->left, depth + 1, nodes, maxDepth);
		if (b->right != b->left) WalkBVH(b->right, depth + 1, nodes, maxDepth);
	}
}
void ref_bvh_stats(void* h, int64_t* outNodes, int32_t* outDepth)
{
	RefScene* rs = (RefScene*)h;
	*outNodes = 0; *outDepth = 0;
	WalkBVH(rs->scene.GetAccelStruct(), 1, outNodes, outDepth);
}

int32_t ref_is_seeded()
{
#ifndef REF_NATIVE_RNG
	return 1;
#else
	return 0;
#endif
}

} // extern "C"
