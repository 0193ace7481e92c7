// Scene / camera host logic: what the reference does in geom/scene.cc,
// render/camera.h and raylib.cc between Raylib_CreateScene and Raylib_FinalizeScene,
// re-stated for a flat device scene.
#include "rl_host.h"

#include <chrono>
#include <string.h>

namespace rl {

// reference render/camera.h:55-78
void Camera::UpdateInternal()
{
	lensRadius = aperture * 0.5f;
	timePeriod = endTime - beginTime;
	w = normalize(origin - lookAt);
	f3 up = F3(0.0f, 1.0f, 0.0f);
	if (dot(w, up) >= 0.9f) up = F3(1.0f, 0.0f, 0.0f);
	u = normalize(cross(up, w));
	v = cross(w, u);
	const float pi_f = (float)3.1415926535897932385;
	float theta = fovY_degrees * pi_f / 180.0f;
	float hh = tanf(theta / 2.0f);
	float hw = aspectWH * hh;
	top_left = origin - (hw * focalDistance * u) - (hh * focalDistance * v) - (focalDistance * w);
	horizontal = 2.0f * hw * focalDistance * u;
	vertical = 2.0f * hh * focalDistance * v;
}

DCamera Camera::ToDevice() const
{
	DCamera d; memset(&d, 0, sizeof(d));
	d.origin[0] = origin.x; d.origin[1] = origin.y; d.origin[2] = origin.z; d.lensRadius = lensRadius;
	d.top_left[0] = top_left.x; d.top_left[1] = top_left.y; d.top_left[2] = top_left.z; d.beginTime = beginTime;
	d.horizontal[0] = horizontal.x; d.horizontal[1] = horizontal.y; d.horizontal[2] = horizontal.z; d.timePeriod = timePeriod;
	d.vertical[0] = vertical.x; d.vertical[1] = vertical.y; d.vertical[2] = vertical.z;
	d.u[0] = u.x; d.u[1] = u.y; d.u[2] = u.z;
	d.v[0] = v.x; d.v[1] = v.y; d.v[2] = v.z;
	return d;
}

// reference geom/scene.cc:6-10
Scene::Scene()
{
	sunIlluminance = F3(0.0f, 0.0f, 0.0f);
	sunDirection = normalize(F3(0.0f, -1.0f, -0.5f));
}

Scene::~Scene()
{
	if (device) DeviceReleaseScene(device);
}

// reference geom/scene.cc:23-31 builds the top BVH over the elements added so far and
// ignores later additions.  Here: concatenate the borrowed models' triangles, offset
// their material / texture indices, build the flat BVH, mark alpha-tested leaves.
void Scene::Finalize()
{
	if (finalized) return;
	finalized = true;
	triangles.clear(); materials.clear(); textures.clear();
	int32_t shapeBase = 0;
	for (OBJModel* m : models) {
		const int32_t matBase = (int32_t)materials.size();
		const int32_t texBase = (int32_t)textures.size();
		for (const std::shared_ptr<Image>& im : m->images) textures.push_back(im);
		for (HostMaterial hm : m->materials) {
			for (int k = 0; k < 5; ++k) if (hm.tex[k] >= 0) hm.tex[k] += texBase;
			materials.push_back(hm);
		}
		for (HostTriangle t : m->triangles) {
			t.material += matBase;
			t.shape += shapeBase;
			triangles.push_back(t);
		}
		shapeBase += m->numShapes;
	}
	// elements added with Raylib_AddSceneElement: spheres, cubes, loose triangles; each brings its material
	spheres.clear(); cubes.clear();
	for (SceneElement* e : elements) {
		const int32_t mat = (int32_t)materials.size();
		HostMaterial hm = e->material->m;
		for (int k = 0; k < 5; ++k) hm.tex[k] = -1;
		materials.push_back(hm);
		if (e->kind == PRIM_SPHERE) { HostSphere s; s.center = e->center; s.radius = e->radius; s.material = mat; spheres.push_back(s); }
		else if (e->kind == PRIM_CUBE) { HostCube c; c.minBounds = e->minBounds; c.maxBounds = e->maxBounds; c.timeStartMove = e->timeStartMove; c.velocity = e->velocity; c.material = mat; cubes.push_back(c); }
		else { HostTriangle t = e->tri; t.material = mat; t.shape = shapeBase++; triangles.push_back(t); }
	}
	hasMovingCubes = false;
	for (const HostCube& c : cubes) if (c.velocity.x != 0.0f || c.velocity.y != 0.0f || c.velocity.z != 0.0f) hasMovingCubes = true;
	if (!BuildAccel(0.0f, 0.0f)) {
		// the reference would run out of memory long before (152-byte triangles + a heap node each); here the limit is the leaf
		// reference's 25-bit slot number.  The scene stays un-finalized: Raylib_Render refuses it with a log line.
		triangles.clear(); triangles.shrink_to_fit(); spheres.clear(); cubes.clear();
		finalized = false;
	}
}

// The flat BVH.  [t0, t1] is the shutter interval the boxes of moving cubes must cover.
// The reference always builds with t0 = t1 = 0 (geom/scene.cc:28) and tests a primitive's own box never, only its
// parent's union box -- so whether it still finds a cube that has moved out of its t = 0 box depends on which sibling
// its random build happened to pair it with.  Here a cube's box covers its whole motion over the camera's shutter
// (Cube::BoundingBox(t0, t1), geom/cube.cc:45-52), i.e. the cube is found wherever it really is.
bool Scene::BuildAccel(float t0, float t1)
{
	accelT0 = t0; accelT1 = t1;
	if (!BVHCapacityOk(triangles.size() + spheres.size() + cubes.size())) {
		Log("Raylib_FinalizeScene: %zu primitives exceed the %u this library's BVH can address; the scene was NOT finalized",
		    triangles.size() + spheres.size() + cubes.size(), (1u << 25) - 1u);
		return false;
	}
	std::vector<PrimRef> prims;
	prims.reserve(triangles.size() + spheres.size() + cubes.size());
	for (size_t i = 0; i < triangles.size(); ++i) {
		const HostTriangle& t = triangles[i];
		PrimRef p; p.mn = fmin3(fmin3(t.v0, t.v1), t.v2); p.mx = fmax3(fmax3(t.v0, t.v1), t.v2); p.kind = PRIM_TRIANGLE; p.index = (uint32_t)i;
		prims.push_back(p);
	}
	for (size_t i = 0; i < spheres.size(); ++i) {   // reference geom/sphere.cc:47-52
		const f3 R = F3(spheres[i].radius, spheres[i].radius, spheres[i].radius);
		PrimRef p; p.mn = spheres[i].center - R; p.mx = spheres[i].center + R; p.kind = PRIM_SPHERE; p.index = (uint32_t)i;
		prims.push_back(p);
	}
	for (size_t i = 0; i < cubes.size(); ++i) {
		const float d0 = t0 - cubes[i].timeStartMove, d1 = t1 - cubes[i].timeStartMove;
		const f3 m0 = cubes[i].velocity * (d0 > 0.0f ? d0 : 0.0f), m1 = cubes[i].velocity * (d1 > 0.0f ? d1 : 0.0f);
		PrimRef p;
		p.mn = fmin3(cubes[i].minBounds + m0, cubes[i].minBounds + m1);
		p.mx = fmax3(cubes[i].maxBounds + m0, cubes[i].maxBounds + m1);
		p.kind = PRIM_CUBE; p.index = (uint32_t)i;
		prims.push_back(p);
	}
	const auto tBuild = std::chrono::steady_clock::now();
	BVHBuildOptions bopt;
	{ const char* e = getenv("RAYLIB_WIDE_GREEDY"); bopt.wideGreedy = e && atoi(e) != 0; }   // read here, once per scene: rl_host.h BVHBuildOptions
	if (const char* e = getenv("RAYLIB_W8_SPLIT")) bopt.splitLeaves8 = atoi(e) != 0;
	if (const char* e = getenv("RAYLIB_W8_TRI_COST")) bopt.triCost8 = (float)atof(e);
	BuildBVH(prims, bvh, bopt);
	const double buildSec = std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild).count();
	// leaves that contain a triangle whose material has an albedo texture run the
	// cut-out test inside traversal (reference geom/triangle.cc:54, material.cc:397-404)
	std::vector<uint8_t> alpha(triangles.size(), 0);
	bool any = false;
	for (size_t i = 0; i < triangles.size(); ++i) {
		const HostMaterial& hm = materials[triangles[i].material];
		if (hm.type == MAT_MICROFACET && hm.tex[0] >= 0) { alpha[i] = 1; any = true; }
	}
	if (any) {
		auto patch = [&](int32_t& ref) {
			if (ref >= 0 || ref == DNODE_EMPTY) return;
			uint32_t code = (uint32_t)~ref, first = code >> 6, count = (code & 7u) + 1;
			if (((code >> 4) & 3u) != PRIM_TRIANGLE) return;
			for (uint32_t k = 0; k < count; ++k) if (alpha[bvh.triOrder[first + k]]) { code |= 8u; break; }
			ref = ~(int32_t)code;
		};
		for (DNode& n : bvh.nodes) { patch(n.left); patch(n.right); }
		for (DNode4& n : bvh.nodes4) for (int k = 0; k < 4; ++k) patch(n.child[k]);
		for (DNode4Q& n : bvh.nodes4q) for (int k = 0; k < 4; ++k) patch(n.child[k]);
		for (DNode4& n : bvh.leafList) for (int k = 0; k < 4; ++k) patch(n.child[k]);
		for (DNode8& n : bvh.nodes8) {   // the 8-wide node names its leaf children's triangles through triBase + leafMask; the flag is a bit per child
			n.alphaMask = 0;
			for (int c = 0; c < 8; ++c) {
				const uint32_t nib = (n.leafMask >> (4 * c)) & 15u;
				if (!nib) continue;
				const uint32_t first = n.triBase + (uint32_t)__builtin_popcount(n.leafMask & ((1u << (4 * c)) - 1u)), count = (uint32_t)__builtin_popcount(nib);
				for (uint32_t k = 0; k < count; ++k) if (alpha[bvh.triOrder[first + k]]) { n.alphaMask |= 1u << c; break; }
			}
		}
	}
	Log("Scene finalized: %u triangles, %u BVH nodes, depth %u, SAH cost %.2f (BVH build %.2f s)",
	    (unsigned)triangles.size(), (unsigned)bvh.nodes.size(), bvh.depth, bvh.sahCost, buildSec);
	if (!bvh.nodes4.empty()) Log("\twide tree: %u BVH4 nodes, worst-case traversal stack %u entries", (unsigned)bvh.nodes4.size(), bvh.stackNeed4);
	if (!bvh.nodes8.empty()) Log("\t8-wide tree: %u nodes, %u levels; expected node steps of a random ray %.1f (4-wide tree: %.1f)", (unsigned)bvh.nodes8.size(), bvh.depth8, bvh.sahNodes8, bvh.sahNodes4);
	if (!bvh.leafList.empty()) {
		unsigned leaves = 0;
		for (const DNode4& nd : bvh.leafList) for (int k = 0; k < 4; ++k) if (nd.child[k] != DNODE_EMPTY) ++leaves;
		Log("\tleaf list: %u leaves in %u records (k_trace walks it instead of the tree)", leaves, (unsigned)bvh.leafList.size());
	}
	return true;
}

// Image2D::PostProcess on the host (reference render/image.cc:44-103): max-luminance
// scan, extended Reinhard on luminance, clamp to white, gamma 1/2.2.
void PostProcessHost(Image& img)
{
	img.SyncHost();
	const size_t len = (size_t)img.width * img.height;
	float maxWhiteLuminance = 1.0f;
	for (size_t i = 0; i < len; ++i) {
		const float* p = &img.rgba[4 * i];
		float L = dot(F3(p[0], p[1], p[2]), F3(0.2126f, 0.7152f, 0.0722f));
		if (maxWhiteLuminance < L) maxWhiteLuminance = L;
	}
	Log("Max white luminance: %f", maxWhiteLuminance);
	for (size_t i = 0; i < len; ++i) {
		float* p = &img.rgba[4 * i];
		f3 rgb = F3(p[0], p[1], p[2]);
		float luminanceOld = dot(rgb, F3(0.2126f, 0.7152f, 0.0722f));
		if (luminanceOld <= 0.0001f) rgb = F3(0.0f, 0.0f, 0.0f);
		else {
			float numerator = luminanceOld * (1.0f + (luminanceOld / (maxWhiteLuminance * maxWhiteLuminance)));
			float luminanceNew = numerator / (1.0f + luminanceOld);
			rgb = rgb * (luminanceNew / luminanceOld);
		}
		rgb = fmin3(F3(1.0f, 1.0f, 1.0f), rgb);
		const float K = 1.0f / 2.2f;
		p[0] = powf(rgb.x, K); p[1] = powf(rgb.y, K); p[2] = powf(rgb.z, K);
	}
}

} // namespace rl
