// HIP kernels + device runtime of the MI355X raylib (gfx950, wave64).
//
// Replaces, on the device, the reference's per-pixel render loop and everything it
// calls (reference render/renderer.cc:62-271, geom/bvh.cc:82-107, geom/aabb.h:14-55,
// geom/triangle.cc:18-58, geom/hit.cc:6-30, render/material.cc, render/brdf.h,
// render/camera.h:44-53, render/texture.cc:30-53, core/random.cc:3-50):
//
//   k_trace    persistent "megakernel": every lane owns one camera sample (a path)
//              at a time and runs one bounce per loop trip; lanes whose path ended
//              are refilled at the top of the loop from a global job counter using a
//              wave64 __ballot + prefix rank (one atomic per wave and trip), so waves
//              stay full while path lengths differ.  Traversal is iterative on ONE
//              flat BVH2 with an LDS stack ([entry][lane], conflict free), ordered
//              near-first with t-shrinking -- it returns the reference's closest hit
//              (min t over all triangles) without the reference's both-children walk.
//   k_resolve  sums a pixel's samples IN SAMPLE ORDER (float addition is not
//              associative; the reference adds s = 0..SPP-1 sequentially,
//              renderer.cc:232-246) and applies the reciprocal-multiply mean.
//   k_aov      the debug render modes (renderer.cc:62-111).
//   k_closest_hit  rays in -> hit records out (tests).
//
// Radiance is folded exactly as the reference's recursion evaluates it
// (renderer.cc:139-151): per bounce the lane stores (reflectance, scatteringPdf,
// pdf, emitted) in a global path stack and, when the path ends, folds from the last
// vertex back to the camera, so every rounding step is the reference's.
#include <hip/hip_runtime.h>

#include "rl_host.h"
// the exact-libm tables (rl_glibc_math.h) in LDS: 640 B per workgroup, filled by rlm_fill_lds_tables() at the top of every kernel that
// evaluates expf / logf / powf.  A microfacet scattering event makes ~16 such look-ups; from constant memory each one is a gather through
// the vector memory pipeline with a full s_waitcnt behind it.
#ifndef RL_MATH_TABLES_GLOBAL
#define RLM_LDS_TABLES 1
__shared__ double rlm_lds_tab[80];
#define RL_MATH_PROLOGUE() rlm::rlm_fill_lds_tables()
#else
#define RL_MATH_PROLOGUE()
#endif
#include "rl_math.h"
#include "raylib_amd_rng.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <limits>
#include <map>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace rl {

#define RL_BLOCK 256
#ifndef RL_POOL_NODEPTR_VGPR
#define RL_POOL_NODEPTR_VGPR 1   /* 298 k-triangle frame 43.0 -> 42.6 ms */
#endif
#ifndef RL_ROOTMISS_RCP
#define RL_ROOTMISS_RCP 1
#endif
#ifndef RL_REFILL_ROUNDS
#define RL_REFILL_ROUNDS 4
#endif
// paths of up to this many vertices fetch all their vertex records before the fold's dependent chain (0: one fetch per step)
#ifndef RL_FOLD_PREFETCH
#define RL_FOLD_PREFETCH 5
#endif
#ifndef RL_FOLD_PREFETCH_POOL
#define RL_FOLD_PREFETCH_POOL RL_FOLD_PREFETCH
#endif

// The wave's lane mask of a predicate, as the exec-masked compare it is.  HIP's __ballot(int) reaches the same builtin through an int: the compiler
// then materialises the bool as 0 / 1 in a VGPR and compares it with zero again (v_cndmask + v_cmp_ne, 8 issue cycles per ballot on kernels that vote
// several times per traversal step).
__device__ __forceinline__ unsigned long long Ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// ---------------------------------------------------------------------------
// device float3 (reference core/vec3.h conventions; see rl_host.h f3)
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 v3s(float s) { return v3(s, s, s); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float t) { return v3(a.x * t, a.y * t, a.z * t); }
__device__ __forceinline__ V3 operator*(float t, V3 a) { return v3(a.x * t, a.y * t, a.z * t); }
__device__ __forceinline__ V3 operator/(V3 a, float t) { return v3(a.x / t, a.y / t, a.z / t); }
__device__ __forceinline__ V3 operator-(V3 a, float t) { return v3(a.x - t, a.y - t, a.z - t); }
__device__ __forceinline__ V3 operator-(float t, V3 a) { return v3(t - a.x, t - a.y, t - a.z); }
__device__ __forceinline__ V3 operator+(V3 a, float t) { return v3(a.x + t, a.y + t, a.z + t); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float absDot(V3 a, V3 b) { return fabsf(a.x * b.x + a.y * b.y + a.z * b.z); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x); }
// (rtm::sqrt_ and rtm::rcp1_ are sqrtf and 1.0f / x bit for bit: rl_math.h)
__device__ __forceinline__ float length(V3 a) { return rtm::sqrt_(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 normalize(V3 a) { float k = rtm::rcp1_(length(a)); return v3(a.x * k, a.y * k, a.z * k); }
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - 2.0f * dot(v, n) * n; }
__device__ __forceinline__ V3 mix(V3 a, V3 b, float t) { return (1.0f - t) * a + t * b; }
__device__ __forceinline__ V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ bool isZero(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }

#define RL_PI 3.14159265359f   /* BRDF::PI, reference render/brdf.h:8 */

// diagnostic build (-DRL_DIAG_TIMELINE=1, RAYLIB_PRINT_STAMPS=1): k_trace's waves record when they start, when they first find the
// job queue empty and when they end (s_memrealtime, 100 MHz), three arrays of 8192 slots behind the counters
#ifdef RL_DIAG_TIMELINE
#define RL_TIMELINE_SLOTS (4 * 8192)   /* start | job list seen empty | end | XCC id */
#define RL_TIMELINE(which) { if (lane == 0 && (gtid >> 6) < 8192u) { countersK[CNT_COUNT + 24 + (which) * 8192 + (gtid >> 6)] = __builtin_amdgcn_s_memrealtime(); if ((which) == 0) countersK[CNT_COUNT + 24 + 3 * 8192 + (gtid >> 6)] = XccId(); } }
#else
#define RL_TIMELINE_SLOTS 0
#define RL_TIMELINE(which)
#endif
#ifdef RL_DIAG_STAMPS
#define RL_DIAG_BIND(c) { (c).diag = nullptr; (c).tLast = 0; (c).tAcc[0] = (c).tAcc[1] = (c).tAcc[2] = (c).tAcc[3] = 0; }
#else
#define RL_DIAG_BIND(c)
#endif
struct Counters {
	uint32_t rays, nodes, tris, shaded, texels, samples, trips;
#ifdef RL_DIAG_STAMPS
	unsigned long long* diag;   // diagnostic build: the global counter array (slots CNT_COUNT + k)
	unsigned long long tLast, tAcc[4];
#endif
};
#ifdef RL_DIAG_STAMPS
#define RL_CSTAMP_BEGIN(c) { __builtin_amdgcn_sched_barrier(0); (c).tLast = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define RL_CSTAMP(c, k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (c).tAcc[k] += now_ - (c).tLast; (c).tLast = now_; __builtin_amdgcn_sched_barrier(0); }
#if RL_DIAG_STAMPS >= 2   /* wave-step against lane-step counts: global atomics in the inner loops, they distort the clock shares */
#define RL_WLSTEP(c, kw, kl) { const unsigned long long em_ = Ballot(1); if ((c).diag && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)em_) - 1u) { atomicAdd(&(c).diag[CNT_COUNT + kw], 1ull); atomicAdd(&(c).diag[CNT_COUNT + kl], (unsigned long long)__popcll(em_)); } }
#else
#define RL_WLSTEP(c, kw, kl)
#endif
#else
#define RL_WLSTEP(c, kw, kl)
#define RL_CSTAMP_BEGIN(c)
#define RL_CSTAMP(c, k)
#endif

// ---------------------------------------------------------------------------
// RNG (include/raylib_amd_rng.h); draws in the reference's program order.
struct Rng { RaylibRngStream s; };
__device__ __forceinline__ float Next(Rng& g) { return raylib_rng_next_float(&g.s); }

// reference core/random.cc:3-23
__device__ __forceinline__ V3 RandomInUnitSphere(Rng& g)
{
	float u1 = Next(g);
	float u2 = Next(g);
	float z = 1.0f - 2.0f * u1;
	float r = rtm::sqrt_(fmaxf(0.0f, 1.0f - z * z));
	float phi = 2.0f * 3.141592f * u2;
	float sn, cs; rtm::sincos_(phi, &sn, &cs);
	return v3(r * cs, r * sn, z);
}
// reference core/random.cc:42-50
__device__ __forceinline__ V3 RandomInUnitDisk(Rng& g)
{
	float u1 = Next(g);
	float u2 = Next(g);
	float r = rtm::sqrt_(u1);
	float theta = 2.0f * 3.14159265358979323846f * u2;
	float sn, cs; rtm::sincos_(theta, &sn, &cs);
	return v3(r * cs, r * sn, 0.0f);
}

// ---------------------------------------------------------------------------
// Texture2D::Sample (reference render/texture.cc:30-53, render/image.h:79-83)
// Out of line (textured scenes only) and fed plain pointers, so that no caller-side struct has its
// address taken (that would push it to scratch).
__device__ __noinline__ float4 TexFetch(const DTexture* textures, const float* texels, int tex, bool srgb, float u, float v)
{
	const DTexture T = textures[tex];
	u = rtm::fmod1_(u); if (u < 0.0f) u += 1.0f;
	v = rtm::fmod1_(v); if (v < 0.0f) v += 1.0f; v = 1.0f - v;
	if (isnan(u) || isinf(u)) u = 0.0f;
	if (isnan(v) || isinf(v)) v = 0.0f;
	int x = (int)((float)(uint32_t)(T.width - 1) * u);
	int y = (int)((float)(uint32_t)(T.height - 1) * v);
	float4 px = ((const float4*)texels)[T.offset + (uint32_t)(y * T.width + x)];
	if (srgb) { px.x = rtm::pow_(px.x, 2.2f); px.y = rtm::pow_(px.y, 2.2f); px.z = rtm::pow_(px.z, 2.2f); px.w = rtm::pow_(px.w, 2.2f); }
	return px;
}
// The texture descriptors (first texel, width, height) of a scene with at most RL_LDS_TEXTURES textures are copied to LDS at the top of the pool kernel
// (RL_TEX_PROLOGUE): a fetch is then descriptor (LDS) -> texel instead of two dependent trips through the vector memory pipeline.  TexFetch takes a generic
// pointer and loads through it with a flat instruction, which serves either address space.  Only in the pool schedule's translation unit: k_trace renders the
// scenes of a few hundred triangles, where a texel fetch is rare, and the Cornell frame was 0.6 % slower with the table in its kernel (12.36 against 12.28 ms).
#ifdef RL_TU_POOL
__shared__ DTexture rl_lds_tex[RL_LDS_TEXTURES];
#define RL_TEX_PROLOGUE(S_) { if ((S_).numTextures <= RL_LDS_TEXTURES) for (int i_ = (int)threadIdx.x; i_ < (S_).numTextures; i_ += (int)blockDim.x) rl_lds_tex[i_] = (S_).textures[i_]; }
__device__ __forceinline__ const DTexture* TexTable(const DSceneView& S) { return (S.numTextures <= RL_LDS_TEXTURES) ? (const DTexture*)rl_lds_tex : S.textures; }
#else
#define RL_TEX_PROLOGUE(S_)
__device__ __forceinline__ const DTexture* TexTable(const DSceneView& S) { return S.textures; }
#endif
__device__ __forceinline__ float4 TexSample(const DSceneView& S, int tex, bool srgb, float u, float v, Counters& c)
{
	c.texels++;
	return TexFetch(TexTable(S), S.texels, tex, srgb, u, v);
}

struct Mat {   // DMaterial in registers
	int type;
	V3 albedo; float roughness, metallic; V3 emissive; float ior; V3 transmission; float fuzz;
	int tex0, tex1, tex2, tex3, tex4;
};
__device__ __forceinline__ Mat LoadMat(const DSceneView& S, int i)
{
	const float4* p = (const float4*)(S.materials + i);
	float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
	Mat m;
	m.type = __float_as_int(a.x); m.albedo = v3(a.y, a.z, a.w);
	m.roughness = b.x; m.metallic = b.y; m.emissive = v3(b.z, b.w, c.x);
	m.ior = c.y; m.transmission = v3(c.z, c.w, d.x); m.fuzz = d.y;
	m.tex0 = __float_as_int(d.z); m.tex1 = __float_as_int(d.w);
	m.tex2 = __float_as_int(e.x); m.tex3 = __float_as_int(e.y); m.tex4 = __float_as_int(e.z);
	return m;
}

// ---------------------------------------------------------------------------
// A scene small enough lives in LDS for the duration of a k_trace workgroup (<= 32 wide nodes, <= 128 triangles, <= 32 materials:
// the Cornell class): the BVH2 root, the float-box wide nodes, both triangle record arrays and the material table, 23 KB at fixed
// offsets (float4 units) so that every access is a ds_read_b128 with an immediate offset.  Every dependent fetch of a bounce --
// three to six node steps, the triangle records, the shading record, the material -- then costs an LDS round trip instead of a trip
// through the vector memory pipeline (TA / L1 / L2), which sixteen waves per CU keep busy with 64-address gathers.
#define RL_LDS_ROOT   0
#define RL_LDS_NODES  4
#ifndef RL_LDS_MAXNODES
#define RL_LDS_MAXNODES 32
#endif
#ifndef RL_LDS_NSTRIDE
#define RL_LDS_NSTRIDE 8   /* float4 per node record (8 = packed) */
#define RL_LDS_TSTRIDE 4   /* float4 per triangle record, both arrays */
#endif
#define RL_LDS_ISECT  (RL_LDS_NODES + RL_LDS_MAXNODES * RL_LDS_NSTRIDE)
#ifndef RL_LDS_MAXTRIS
#define RL_LDS_MAXTRIS 128
#endif
#define RL_LDS_SHADE  (RL_LDS_ISECT + RL_LDS_MAXTRIS * RL_LDS_TSTRIDE)
#define RL_LDS_MATS   (RL_LDS_SHADE + RL_LDS_MAXTRIS * RL_LDS_TSTRIDE)
#ifndef RL_LDS_MAXMATS
#define RL_LDS_MAXMATS 32
#endif
#define RL_LDS_TOTAL  (RL_LDS_MATS + RL_LDS_MAXMATS * 5)
// The leaf-list kernel (LDS == 2) has its own layout: six records of leaf boxes instead of a tree, at most 108 triangles, and an intersection
// record of SIX float4 that holds what the triangle test would otherwise recompute per test -- the edges u = v1 - v0, v = v2 - v0 (triangle.cc:30-31)
// and the triangle's own box (the candidate rule's) -- computed once per workgroup when the scene is copied in, with the same operations.
template <int LDS> struct LdsAt {
	static constexpr int NODES = RL_LDS_NODES;
	static constexpr int MAXNODES = LDS == 2 ? RL_LEAFLIST_RECORDS : RL_LDS_MAXNODES;
	static constexpr int TRI = LDS == 2 ? 6 : RL_LDS_TSTRIDE;                 // float4 per intersection record
	static constexpr int MAXTRIS = LDS == 2 ? RL_LEAFLIST_MAXTRIS : RL_LDS_MAXTRIS;
	static constexpr int ISECT = NODES + MAXNODES * RL_LDS_NSTRIDE;
	static constexpr int SHADE = ISECT + MAXTRIS * TRI;
	static constexpr int MATS = SHADE + MAXTRIS * RL_LDS_TSTRIDE;
	static constexpr int TOTAL = MATS + RL_LDS_MAXMATS * 5;
};

__device__ __forceinline__ Mat MatFrom(const float4* p)
{
	float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
	Mat m;
	m.type = __float_as_int(a.x); m.albedo = v3(a.y, a.z, a.w);
	m.roughness = b.x; m.metallic = b.y; m.emissive = v3(b.z, b.w, c.x);
	m.ior = c.y; m.transmission = v3(c.z, c.w, d.x); m.fuzz = d.y;
	m.tex0 = __float_as_int(d.z); m.tex1 = __float_as_int(d.w);
	m.tex2 = __float_as_int(e.x); m.tex3 = __float_as_int(e.y); m.tex4 = __float_as_int(e.z);
	return m;
}

// ---------------------------------------------------------------------------
// Closest hit on the flat BVH2.
struct HitRec { float t, a, b; int tri; };   // tri: triangle slot, or (kind << 28) | index for sphere (1) / cube (2, with the face in a)

struct Tri { V3 v0, n, v1, v2, u, v; float uv, uu, vv, denom, rden; };
__device__ __forceinline__ Tri TriFrom(const float4* p)
{
	float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
	Tri t;
	t.v0 = v3(q0.x, q0.y, q0.z); t.n = v3(q0.w, q1.x, q1.y);
	t.v1 = v3(q1.z, q1.w, q2.x); t.v2 = v3(q2.y, q2.z, q2.w);
	t.u = t.v1 - t.v0; t.v = t.v2 - t.v0;   // geom/triangle.cc:30-31
	t.uv = q3.x; t.uu = q3.y; t.vv = q3.z; t.rden = q3.w;
	t.denom = t.uv * t.uv - t.uu * t.vv;    // geom/triangle.cc:39-41, the host's own three operations (rl_runtime.inl UploadScene): the record's slot holds 1 / denom
	return t;
}
__device__ __forceinline__ Tri LoadTri(const DSceneView& S, int i) { return TriFrom((const float4*)(S.isect + i)); }

struct Shade { V3 n0, n1, n2; float s0, t0, s1, t1, s2, t2; int material; };
__device__ __forceinline__ Shade ShadeFrom(const float4* p)
{
	float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
	Shade s;
	s.n0 = v3(q0.x, q0.y, q0.z); s.n1 = v3(q0.w, q1.x, q1.y); s.n2 = v3(q1.z, q1.w, q2.x);
	s.s0 = q2.y; s.t0 = q2.z; s.s1 = q2.w; s.t1 = q3.x; s.s2 = q3.y; s.t2 = q3.z;
	s.material = __float_as_int(q3.w);
	return s;
}
__device__ __forceinline__ Shade LoadShade(const DSceneView& S, int i) { return ShadeFrom((const float4*)(S.shade + i)); }

// MicrofacetMaterial::AlphaTest for a candidate (reference render/material.cc:397-404 via geom/triangle.cc:48-54).
// Returns bit 0 = passes, bit 1 = a texel was fetched.  Out of line: only leaves flagged as textured reach it.
__device__ __noinline__ int AlphaTestCandidateNI(const DTriShade* shade, const int32_t* alphaTex, const DMaterial* materials, const DTexture* textures,
                                                 const float* texels, int tri, float a, float b)
{
	const float4* p = (const float4*)(shade + tri);
	// the texture comes from the per-triangle table (rl_runtime.inl UploadScene), fetched beside the triangle's UVs: one dependent load fewer than through the material
	int tex = alphaTex ? alphaTex[tri] : 0;
	const float4 q2 = p[2], q3 = p[3];
	const float s0 = q2.y, t0 = q2.z, s1 = q2.w, t1 = q3.x, s2 = q3.y, t2 = q3.z;
	if (!alphaTex) {
		const DMaterial* M = materials + __float_as_int(q3.w);
		tex = (M->type == MAT_MICROFACET) ? M->tex[0] : -1;
	}
	if (tex < 0) return 1;
	float U = (1 - a - b) * s0 + a * s1 + b * s2;
	float V = (1 - a - b) * t0 + a * t1 + b * t2;
	float4 px = TexFetch(textures, texels, tex, false, U, V);   // the pow(2.2) copy made at upload
	return (px.w >= 0.5f ? 1 : 0) | 2;
}
__device__ __forceinline__ bool AlphaTestCandidate(const DSceneView& S, int tri, float a, float b, Counters& c)
{
	const int r = AlphaTestCandidateNI(S.shade, S.alphaTex, S.materials, TexTable(S), S.texels, tri, a, b);
	c.shaded++;
	if (r & 2) c.texels++;
	return (r & 1) != 0;
}

// Relative slack of every box test of a traversal (and of the candidate rule below): far above float rounding of the slab
// arithmetic (a few ulp), far below anything visible.
#define RL_BOX_WIDEN 1.00001f
#define RL_CANDIDATE_SLACK 1.000009f   /* a little less than the boxes' slack: the pool schedule's v_rcp_f32 reciprocals may move a box entry by an ulp */

// A candidate that passed the triangle test counts only if the ray also passes the reference's own box test
// (geom/aabb.h:39-54, unwidened, t_max = FLT_MAX) on the triangle's exact AABB.  Why: the barycentric test accepts points a few
// ulp -- on slivers far more -- outside the triangle, i.e. outside every box around it; whether such a candidate is ever REACHED then
// depends on which boxes a traversal happens to test (BVH2 or BVH4, widened by 3 or 6 ulp, the reference's random tree).  With
// this rule the set of accepted hits is a property of the ray and the triangle alone: every schedule and tree width returns the
// same hit, and since every box of the reference's tree contains this AABB (and rounding is monotone) the reference accepts
// whatever is accepted here.  (What it accepts beyond that -- a hit outside the triangle's own box but inside its random
// parent's -- is tree-dependent on its side; the oracle counts those events so that tests can tell them from real mismatches.)
__device__ __forceinline__ bool OwnBoxPassBox(V3 mn, V3 mx, V3 o, V3 inv /* exact 1/d */, float tMin, float t);
__device__ __forceinline__ bool OwnBoxPass(V3 a, V3 b, V3 c, V3 o, V3 inv /* exact 1/d */, float tMin, float t)
{
	const V3 mn = v3(fminf(fminf(a.x, b.x), c.x), fminf(fminf(a.y, b.y), c.y), fminf(fminf(a.z, b.z), c.z));
	const V3 mx = v3(fmaxf(fmaxf(a.x, b.x), c.x), fmaxf(fmaxf(a.y, b.y), c.y), fmaxf(fmaxf(a.z, b.z), c.z));
	return OwnBoxPassBox(mn, mx, o, inv, tMin, t);
}
// the same with the box in hand (the leaf-list kernel's six-float4 triangle record keeps it in float4 4 and 5)
__device__ __forceinline__ bool OwnBoxPassMnMx(const float4* rec, V3 o, V3 inv, float tMin, float t)
{
	const float4 q4 = rec[4], q5 = rec[5];
	return OwnBoxPassBox(v3(q4.x, q4.y, q4.z), v3(q4.w, q5.x, q5.y), o, inv, tMin, t);
}
__device__ __forceinline__ bool OwnBoxPassBox(V3 mn, V3 mx, V3 o, V3 inv /* exact 1/d */, float tMin, float t)
{
	// (lo = t0 > lo ? t0 : lo and hi = t1 < hi ? t1 : hi -- a NaN keeps the old bound -- are fmaxf(lo, t0) and fminf(hi, t1), one v_max / v_min each
	// instead of a compare and a select; the sign of a zero, the one thing the two forms may disagree on, plays no part in the comparisons below)
	float lo = tMin, hi = FLT_MAX;
	{ float t0 = (mn.x - o.x) * inv.x, t1 = (mx.x - o.x) * inv.x; if (inv.x < 0.0f) { const float q = t0; t0 = t1; t1 = q; } lo = fmaxf(lo, t0); hi = fminf(hi, t1); }
	bool ok = !(hi < lo);
	{ float t0 = (mn.y - o.y) * inv.y, t1 = (mx.y - o.y) * inv.y; if (inv.y < 0.0f) { const float q = t0; t0 = t1; t1 = q; } lo = fmaxf(lo, t0); hi = fminf(hi, t1); }
	ok = ok && !(hi < lo);
	{ float t0 = (mn.z - o.z) * inv.z, t1 = (mx.z - o.z) * inv.z; if (inv.z < 0.0f) { const float q = t0; t0 = t1; t1 = q; } lo = fmaxf(lo, t0); hi = fminf(hi, t1); }
	// ... and the candidate's t must not lie before the ray enters that box (by more than the slack the box tests are
	// widened by): then "this box starts beyond the best hit so far" implies "nothing in it is closer", whatever the order
	return ok && !(hi < lo) && t * RL_CANDIDATE_SLACK >= lo;
}

// Slab test of one child box against [tMin, tMax] (reference geom/aabb.h:39-54:
// same products (bound - o) * invD, same "swap if invD < 0", NaN keeps the old
// bound).  tMax is widened by 2 ulp so the test stays conservative.
__device__ __forceinline__ bool Slab(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                     V3 o, V3 inv, bool nx, bool ny, bool nz, float tMin, float tMax, float& tNear, const float widen = RL_BOX_WIDEN)
{
	float tn = tMin, tf = tMax;
	float a0 = ((nx ? mxx : mnx) - o.x) * inv.x, a1 = ((nx ? mnx : mxx) - o.x) * inv.x;
	tn = fmaxf(tn, a0); tf = fminf(tf, a1);
	float b0 = ((ny ? mxy : mny) - o.y) * inv.y, b1 = ((ny ? mny : mxy) - o.y) * inv.y;
	tn = fmaxf(tn, b0); tf = fminf(tf, b1);
	float c0 = ((nz ? mxz : mnz) - o.z) * inv.z, c1 = ((nz ? mnz : mxz) - o.z) * inv.z;
	tn = fmaxf(tn, c0); tf = fminf(tf, c1);
	tNear = tn;
	return !(tf * widen < tn);
}

// v_rcp_f32 (1 ulp) is enough for the slab test's 1/d when the test is widened to 6 ulp (RL_POOL_WIDEN) instead of 3:
// Slab() is only asked to be conservative.  0 -> inf and the sign of a zero survive, as with the division.
__device__ __forceinline__ float FastRcp(float x) { return __builtin_amdgcn_rcpf(x); }

// First traversal step only: true when the ray misses both child boxes of the root node.
template <int LDS = 0>
__device__ __forceinline__ bool RootMiss(const DSceneView& S, V3 o, V3 d, float tMin, const float4* sm = nullptr)
{
	// a filter: "true" only has to imply that a traversal finds nothing.  v_rcp_f32 reciprocals (1 ulp; 8 issue cycles each against the 36 of an
	// IEEE division) under the 1e-5 widening of every box test here, as in the pool schedule's slab tests.  0 -> inf and the sign of a zero survive.
#if RL_ROOTMISS_RCP
	const V3 inv = v3(FastRcp(d.x), FastRcp(d.y), FastRcp(d.z));
#else
	const V3 inv = v3(rtm::rcp1_(d.x), rtm::rcp1_(d.y), rtm::rcp1_(d.z));
#endif
	const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
	const float4* np = LDS ? sm + RL_LDS_ROOT : (const float4*)(S.nodes);
	const float4 q0 = np[0], q1 = np[1], q2 = np[2];
	const int4 k = ((const int4*)np)[3];
	float tl, tr;
	bool hl = Slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, nx, ny, nz, tMin, FLT_MAX, tl);
	bool hr = Slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, nx, ny, nz, tMin, FLT_MAX, tr);
	hl = hl && (k.x != DNODE_EMPTY);
	hr = hr && (k.y != DNODE_EMPTY);
	return !(hl || hr);
}

// stk: this lane's column of the LDS stack; entry k at stk[k * RL_BLOCK].
// Sphere::Hit (reference geom/sphere.cc:3-45): open interval (t_min, t_max); the near root if it is inside, else the
// far root.  t_max is the current best t (the reference compares all hits afterwards; same closest hit).
// Returns t, or NaN for a miss (results by value: an out-parameter of an out-of-line function would live in scratch).
__device__ __noinline__ float SphereHit(const DSphere* spheres, int index, V3 o, V3 d, float t_min, float tBest)
{
	const float4 q = ((const float4*)(spheres + index))[0];
	const V3 center = v3(q.x, q.y, q.z); const float radius = q.w;
	V3 oc = o - center;
	float a = dot(d, d);
	float b = dot(oc, d);
	float c = dot(oc, oc) - radius * radius;
	float D = b * b - a * c;
	if (D > 0.0f) {
		float temp = (-b - rtm::sqrt_(b * b - a * c)) / a;
		if (t_min < temp && temp < FLT_MAX) return (temp < tBest) ? temp : NAN;   // the reference takes this root and compares later
		temp = (-b + rtm::sqrt_(b * b - a * c)) / a;
		if (t_min < temp && temp < FLT_MAX) return (temp < tBest) ? temp : NAN;
	}
	return NAN;
}
// Cube::Hit (reference geom/cube.cc:3-43): slab box moving with velocity * max(0, rayTime - timeStartMove); closed
// interval [t_min, t_max]; entry face by the reference's float == chain (outFace 0..5 = -x +x -y +y -z +z, 6 = none matched).
// Returns (t, face as int bits), t = NaN for a miss.
__device__ __noinline__ float2 CubeHit(const DCube* cubes, int index, V3 o, V3 d, float rayTime, float t_min, float tBest)
{
	const float4* p = (const float4*)(cubes + index);
	const float4 q0 = p[0], q1 = p[1], q2 = p[2];
	const V3 velocity = v3(q2.x, q2.y, q2.z);
	const V3 movement = velocity * fmaxf(0.0f, rayTime - q0.w);
	const V3 mn = v3(q0.x, q0.y, q0.z) + movement, mx = v3(q1.x, q1.y, q1.z) + movement;
	const float t1 = (mn.x - o.x) / d.x, t2 = (mx.x - o.x) / d.x;
	const float t3 = (mn.y - o.y) / d.y, t4 = (mx.y - o.y) / d.y;
	const float t5 = (mn.z - o.z) / d.z, t6 = (mx.z - o.z) / d.z;
	// std::max(a, b) = (a < b) ? b : a; std::min(a, b) = (b < a) ? b : a
	#define RL_STDMAX(a, b) (((a) < (b)) ? (b) : (a))
	#define RL_STDMIN(a, b) (((b) < (a)) ? (b) : (a))
	const float mnx = RL_STDMIN(t1, t2), mny = RL_STDMIN(t3, t4), mnz = RL_STDMIN(t5, t6);
	const float mxx = RL_STDMAX(t1, t2), mxy = RL_STDMAX(t3, t4), mxz = RL_STDMAX(t5, t6);
	const float m12 = RL_STDMAX(mnx, mny); const float t7 = RL_STDMAX(m12, mnz);
	const float n12 = RL_STDMIN(mxx, mxy); const float t8 = RL_STDMIN(n12, mxz);
	#undef RL_STDMAX
	#undef RL_STDMIN
	if (t8 < 0 || t7 > t8) return make_float2(NAN, 0.0f);
	if (t_min <= t7 && t7 <= FLT_MAX && t7 < tBest) {
		const int face = (t7 == t1) ? 0 : (t7 == t2) ? 1 : (t7 == t3) ? 2 : (t7 == t4) ? 3 : (t7 == t5) ? 4 : (t7 == t6) ? 5 : 6;
		return make_float2(t7, __int_as_float(face));
	}
	return make_float2(NAN, 0.0f);
}

// The barycentric coordinates of a plane hit and their test, reference geom/triangle.cc:41-47:  pa = X / denom, pb = Y / denom, inside <=> 0 <= pa, 0 <= pb,
// pa + pb <= 1.  Two IEEE divisions are 72 of the ~380 issue cycles of a triangle step, and the divisor is a constant of the triangle: with rden = RN(1 / denom)
// from the record, rtm::div_by_ gives the same two quotients in 12 (all 2^46 significand pairs checked: tools/verify_fastdiv.hip).  Its conditions -- the ones
// v_div_scale tests -- are met like this:
//   * S.fastBary (host, rl_runtime.inl UploadScene): every triangle of the scene has denom == 0 or NaN (rden = NaN: both quotients NaN, "outside", as X / 0 and
//     X / NaN make it) or 2^-62 <= |denom| <= 2^125; a scene with any other divisor takes the divisions (a uniform branch);
//   * a quotient of at least 2^-38 then has |X| > 2^-101 (div_by_ wants 2^-102): exact.  Anything smaller -- tiny, zero (whose sign the short form may get wrong), negative by less
//     than that -- may be off in the last place, which cannot change "pa + pb <= 1" (a term below 2^-38 moves a sum near 1 by less than a thousandth of its
//     half-ulp), so: outside by more than 2^-38 is outside, inside by more than 2^-38 on both is inside, and the band between takes the divisions and the
//     reference's own test.  (A ray through a vertex or along an edge; tests/test_gpu_parity.py aims rays there.)
#ifndef RL_FAST_BARY
#define RL_FAST_BARY 1
#endif
__device__ __forceinline__ bool Barycentric(bool fast, float X, float Y, float denom, float rden, float& pa, float& pb)
{
#if RL_FAST_BARY
	if (fast) {
		pa = rtm::div_by_(X, denom, rden); pb = rtm::div_by_(Y, denom, rden);
		const float eps = 3.637978807091713e-12f;   // 2^-38
		const float m = __builtin_fminf(pa, pb), sum = pa + pb;   // (a NaN quotient: the sum is NaN)
		if (!(sum <= 1.0f && m >= -eps)) return false;
		if (m >= eps) return true;
	}
#endif
	pa = X / denom; pb = Y / denom;
	return 0.0f <= pa && 0.0f <= pb && pa + pb <= 1.0f;
}

// diagnostic build only: count wave-level steps (first active lane adds 1) next to the lane-level counters
#if defined(RL_DIAG_STAMPS) && RL_DIAG_STAMPS >= 2
#define RL_WSTEP(k) { const unsigned long long em_ = Ballot(1); if (c.diag && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)em_) - 1u) atomicAdd(&c.diag[CNT_COUNT + k], 1ull); }
#else
#define RL_WSTEP(k)
#endif

// "while-while" traversal: every lane first descends through inner nodes until it holds a leaf (cheap steps:
// one 64-byte record, two slab tests), THEN the wave intersects leaves together.  With a single
// "if inner else leaf" loop a wave pays node + leaf cost on every trip as soon as one lane is at a leaf, and
// the ~4x dearer triangle code ran with a handful of lanes (measured: 14 % VALU lane utilisation on the
// 298 k-triangle scene).
template <int STACK, bool ANYHIT, bool PRIMS>
__device__ __forceinline__ bool Traverse(const DSceneView& S, V3 o, V3 d, float rayTime, float tMin, HitRec& best, int* stk, Counters& c)
{
	c.rays++;
	const V3 inv = v3(rtm::rcp1_(d.x), rtm::rcp1_(d.y), rtm::rcp1_(d.z));
	const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
	best.t = INFINITY; best.tri = -1; best.a = 0.0f; best.b = 0.0f;
	int sp = 0;
	int cur = 0;                  // root is an inner node
	const int DONE = 0x7fffffff;  // not a node index (nodes < 2^31 - 1), not negative
	for (;;) {
		// ---- descend: inner nodes until this lane holds a leaf or has nothing left ----
		while (cur >= 0 && cur != DONE) {
			RL_WSTEP(4);
			const float4* np = (const float4*)(S.nodes + cur);
			const float4 q0 = np[0], q1 = np[1], q2 = np[2];
			const int4 k = ((const int4*)np)[3];
			c.nodes++;
			float tl, tr;
			const float tmx = fminf(best.t, FLT_MAX);
			bool hl = Slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, nx, ny, nz, tMin, tmx, tl);
			bool hr = Slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, nx, ny, nz, tMin, tmx, tr);
			hl = hl && (k.x != DNODE_EMPTY);
			hr = hr && (k.y != DNODE_EMPTY);
			if (hl && hr) {
				const bool leftFirst = tl <= tr;
				const int nearC = leftFirst ? k.x : k.y, farC = leftFirst ? k.y : k.x;
				if (sp < STACK) { stk[sp * RL_BLOCK] = farC; ++sp; }
				cur = nearC;
			} else if (hl) cur = k.x;
			else if (hr) cur = k.y;
			else if (sp == 0) cur = DONE;
			else { --sp; cur = stk[sp * RL_BLOCK]; }
		}
		if (cur == DONE) break;
		// ---- leaf: <= 4 triangles stored back to back, or one analytic primitive ----
		{
			RL_WSTEP(6);
			const uint32_t code = (uint32_t)~cur;
			const int first = (int)(code >> 6);
			const int count = (int)(code & 7u) + 1;
			const bool alpha = (code & 8u) != 0;
			const uint32_t kind = (code >> 4) & 3u;
			if (!PRIMS || kind == 0u) {
				for (int i = 0; i < count; ++i) {
					RL_WSTEP(5);
					const Tri T = LoadTri(S, first + i);
					c.tris++;
					// reference geom/triangle.cc:22-27
					const float t = dot((T.v0 - o), T.n) / dot(d, T.n);
					// closer, or exactly as far with a lower slot: which of two surfaces at the same t wins must not depend on the
					// order a traversal happens to test them in (the reference's answer there depends on its random tree, SURVEY A)
					if (!(t >= tMin && t <= FLT_MAX && (t < best.t || (t == best.t && first + i < best.tri)))) continue;
					const V3 p = o + t * d;
					const V3 w = p - T.v0;
					const float wv = dot(w, T.v), wu = dot(w, T.u);
					float pa, pb;
					if (Barycentric(S.fastBary != 0, T.uv * wv - T.vv * wu, T.uv * wu - T.uu * wv, T.denom, T.rden, pa, pb) && OwnBoxPass(T.v0, T.v1, T.v2, o, inv, tMin, t)) {
						if (alpha && !AlphaTestCandidate(S, first + i, pa, pb, c)) continue;
						best.t = t; best.a = pa; best.b = pb; best.tri = first + i;
						if (ANYHIT) return true;
					}
				}
			} else {
				c.tris++;
				float2 r;
				if (kind == 1u) r = make_float2(SphereHit(S.spheres, first, o, d, tMin, best.t), 0.0f);
				else r = CubeHit(S.cubes, first, o, d, rayTime, tMin, best.t);
				if (r.x == r.x) {   // not NaN: a hit
					best.t = r.x; best.a = r.y; best.b = 0.0f; best.tri = (int)((kind << 28) | (uint32_t)first);
					if (ANYHIT) return true;
				}
			}
		}
		if (sp == 0) break;
		--sp;
		cur = stk[sp * RL_BLOCK];
	}
	return best.tri >= 0;
}

// ---- one step on the wide tree: entry distances t0..t3 (INFINITY: not entered) of the four children of S.nodes4[cur] ----
// RL_Q4 (default): the 64-byte grid node (DNode4Q).  The planes are never decoded: with A = step * inv and B = (origin - o) * inv
// per axis, plane q's parameter is fma(q, A, B) -- one v_cvt_f32_ubyte and one v_fma per plane, four 16-byte loads per lane instead
// of seven.  The fused form rounds differently from the reference's (bound - o) * inv, by at most (|B| + 255 |A|) * 2^-23 in
// absolute terms (cancellation when the ray starts inside the node); four times that bound widens every slab -- near planes earlier,
// far planes later.  A box test only has to be conservative (the candidate rule decides what counts as a hit), so the image does
// not change.  A zero direction component (inv = +-inf) would turn the fused form into inf - inf: inv is clamped to +-1e30 for the
// box tests, which keeps the "no constraint while the origin lies between the planes" meaning and errs towards visiting.
__device__ __forceinline__ V3 ClampInv(V3 inv)
{
	// only infinities: a finite reciprocal, however large, scales its axis' parameters exactly as the reference's arithmetic does
	return v3(isinf(inv.x) ? copysignf(1e30f, inv.x) : inv.x, isinf(inv.y) ? copysignf(1e30f, inv.y) : inv.y, isinf(inv.z) ? copysignf(1e30f, inv.z) : inv.z);
}
typedef float rl_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t rl_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 GLoadF4(const void* p, int i) { const rl_v4f v = ((const __attribute__((address_space(1))) rl_v4f*)p)[i]; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 GLoadU4(const void* p, int i) { const rl_v4u v = ((const __attribute__((address_space(1))) rl_v4u*)p)[i]; return make_uint4(v.x, v.y, v.z, v.w); }
#define RL_WIDE_STEP_Q(S_, cur_, o_, inv_, nx_, ny_, nz_, tMin_, tmx_, widen_, t0, t1, t2, t3, ch) \
	/* (the loads spell the global address space out: the pool kernel keeps the base in a VGPR pair behind an empty asm statement, which hides where it */ \
	/*  points -- and a flat_load counts against the LDS counter as well and waits for both) */ \
	const DNode4Q* np_ = (S_).nodes4 + (cur_); \
	const float4 h0_ = GLoadF4(np_, 0); const uint4 l_ = GLoadU4(np_, 1); const uint4 u_ = GLoadU4(np_, 2); \
	const uint4 chu_ = GLoadU4(np_, 3); const int4 ch = make_int4((int)chu_.x, (int)chu_.y, (int)chu_.z, (int)chu_.w); \
	const float Ax_ = h0_.w * (inv_).x, Ay_ = __uint_as_float(l_.w) * (inv_).y, Az_ = __uint_as_float(u_.w) * (inv_).z; \
	const float Bx_ = (h0_.x - (o_).x) * (inv_).x, By_ = (h0_.y - (o_).y) * (inv_).y, Bz_ = (h0_.z - (o_).z) * (inv_).z; \
	/* (|B| + 255 |A|) * 2^-21 as |A * c1| + |B * c2|: two multiplies by literals and an add with |.| modifiers, 2 issue cycles each -- as an fma with 255 the */ \
	/* constant sat in an SGPR (the three-operand encoding takes no literal) next to the |.| modifiers, and an SGPR operand makes it 4 */ \
	const float Ex_ = fabsf(Ax_ * 1.21593475e-4f) + fabsf(Bx_ * 4.76837158e-7f), Ey_ = fabsf(Ay_ * 1.21593475e-4f) + fabsf(By_ * 4.76837158e-7f), Ez_ = fabsf(Az_ * 1.21593475e-4f) + fabsf(Bz_ * 4.76837158e-7f); \
	const float Bnx_ = Bx_ - Ex_, Bfx_ = Bx_ + Ex_, Bny_ = By_ - Ey_, Bfy_ = By_ + Ey_, Bnz_ = Bz_ - Ez_, Bfz_ = Bz_ + Ez_; \
	const uint32_t nX_ = (nx_) ? u_.x : l_.x, fX_ = (nx_) ? l_.x : u_.x, nY_ = (ny_) ? u_.y : l_.y, fY_ = (ny_) ? l_.y : u_.y, nZ_ = (nz_) ? u_.z : l_.z, fZ_ = (nz_) ? l_.z : u_.z; \
	const float tMinL_ = (tMin_), tmxL_ = (tmx_), widenL_ = (widen_); \
	float t0, t1, t2, t3; \
	RL_QSLAB(0, t0) RL_QSLAB(8, t1) RL_QSLAB(16, t2) RL_QSLAB(24, t3)
#define RL_QSLAB(sh, tk) { \
	float tn = tMinL_, tf = tmxL_; \
	tn = fmaxf(tn, __builtin_fmaf((float)((nX_ >> sh) & 0xffu), Ax_, Bnx_)); tf = fminf(tf, __builtin_fmaf((float)((fX_ >> sh) & 0xffu), Ax_, Bfx_)); \
	tn = fmaxf(tn, __builtin_fmaf((float)((nY_ >> sh) & 0xffu), Ay_, Bny_)); tf = fminf(tf, __builtin_fmaf((float)((fY_ >> sh) & 0xffu), Ay_, Bfy_)); \
	tn = fmaxf(tn, __builtin_fmaf((float)((nZ_ >> sh) & 0xffu), Az_, Bnz_)); tf = fminf(tf, __builtin_fmaf((float)((fZ_ >> sh) & 0xffu), Az_, Bfz_)); \
	tk = (tf * widenL_ < tn) ? INFINITY : tn; }
#define RL_WIDE_STEP_F(np_expr, o_, inv_, nx_, ny_, nz_, tMin_, tmx_, widen_, t0, t1, t2, t3, ch) \
	const float4* np_ = (np_expr); \
	const float4 lox_ = np_[0], loy_ = np_[1], loz_ = np_[2], hix_ = np_[3], hiy_ = np_[4], hiz_ = np_[5]; \
	const int4 ch = ((const int4*)np_)[6]; \
	const float4 nX_ = (nx_) ? hix_ : lox_, fX_ = (nx_) ? lox_ : hix_; \
	const float4 nY_ = (ny_) ? hiy_ : loy_, fY_ = (ny_) ? loy_ : hiy_; \
	const float4 nZ_ = (nz_) ? hiz_ : loz_, fZ_ = (nz_) ? loz_ : hiz_; \
	const float tMinL_ = (tMin_), tmxL_ = (tmx_), widenL_ = (widen_); const V3 oL_ = (o_), invL_ = (inv_); \
	float t0, t1, t2, t3; \
	RL_FSLAB(x, t0) RL_FSLAB(y, t1) RL_FSLAB(z, t2) RL_FSLAB(w, t3)
#define RL_FSLAB(k, tk) { \
	float tn = tMinL_, tf = tmxL_; \
	tn = fmaxf(tn, (nX_.k - oL_.x) * invL_.x); tf = fminf(tf, (fX_.k - oL_.x) * invL_.x); \
	tn = fmaxf(tn, (nY_.k - oL_.y) * invL_.y); tf = fminf(tf, (fY_.k - oL_.y) * invL_.y); \
	tn = fmaxf(tn, (nZ_.k - oL_.z) * invL_.z); tf = fminf(tf, (fZ_.k - oL_.z) * invL_.z); \
	tk = (tf * widenL_ < tn) ? INFINITY : tn; }

// A scene of at most 4 * RL_LEAFLIST_RECORDS leaves (rl_bvh.cc "the leaf list"), resident in LDS: no tree.  Every lane tests the box of every
// leaf, four to a record, in lockstep -- the same code on the same records for all 64 rays, so the wave pays for 1 walk, not for the union
// of 64 -- and keeps what it hit as sortable keys: the entry distance with the slot number in the 5 low mantissa bits, i.e. rounded DOWN by
// at most 31 ulp (nearer than the truth, so the cut below only comes later; a negative entry distance, possible with a negative rayTMin,
// counts as 0, and the cut is then not taken at all).  Then it visits its leaves nearest first and stops at the first one that starts behind the best hit -- the order and the cut
// of a tree walk.  The candidates are every leaf whose box the ray meets: a superset of those a tree walk opens, and with the candidate rule
// and the tie rule of the triangle test the result does not depend on which superset is tested in which order.  The cut is safe for the
// same reason every widened box test here is: an accepted hit has t * RL_CANDIDATE_SLACK >= the entry into its triangle's own box (OwnBoxPass),
// which lies inside the leaf's box, and RL_BOX_WIDEN exceeds RL_CANDIDATE_SLACK by 1e-6 -- four times the rounding of either side.
// Measured on the Cornell frame: DESIGN.md section 2.
// (Round 3, measured and not kept: the candidate rule applied once, to the winner of the search, instead of to every candidate that passes the barycentric
// test, with an out-of-line walk that applies it per candidate for the lane whose winner fails it.  Sound -- the search finds the nearest of a larger set, and
// a winner that passes the rule is the nearest of the smaller one too -- and 0.5 % faster, but the call made the register allocator keep the 24 keys in
// scratch memory: 28 GB of spill traffic per frame, three times everything else the kernel moves.)
template <bool ANYHIT>
__device__ __forceinline__ bool TraverseLeafList(const DSceneView& S, V3 o, V3 d, float tMin, HitRec& best, Counters& c, const float4* sm)
{
	c.rays++;
	const V3 invb = v3(rtm::rcp1_(d.x), rtm::rcp1_(d.y), rtm::rcp1_(d.z));
	const bool nx = invb.x < 0.0f, ny = invb.y < 0.0f, nz = invb.z < 0.0f;
	best.t = INFINITY; best.tri = -1; best.a = 0.0f; best.b = 0.0f;
	uint32_t key[4 * RL_LEAFLIST_RECORDS];
	// The box test of the list is a filter, not the reference's test (that one is the candidate rule of the triangle test, on the
	// triangle's own box): it only has to let through every leaf the exact test would.  So the planes are one fma each,
	// t = plane * inv + c with c = -(o * inv), instead of (plane - o) * inv; c's rounding error, |c| * 2^-24, which the exact form does not
	// have when plane ~ o, is covered four times over by moving c outwards by |c| * 2^-22 (near planes down, far planes up), and the
	// relative errors by the same "tf * widen < tn" as every other box test here.  An infinite inv (a zero in d) turns the axis's terms
	// into NaN or into the harmless infinity, which max / min ignore: the axis then simply does not cull.  And the near / far plane of
	// an axis is picked by ADDRESS (the record holds lo.x lo.y lo.z hi.x hi.y hi.z, 16 bytes each) instead of by 24 selects per record.
	const V3 cc = v3(-(o.x * invb.x), -(o.y * invb.y), -(o.z * invb.z));
	const V3 ce = v3(fabsf(cc.x) * 2.3841858e-7f, fabsf(cc.y) * 2.3841858e-7f, fabsf(cc.z) * 2.3841858e-7f);
	const V3 cn = cc - ce, cf = cc + ce;
	const char* recs = (const char*)(sm + LdsAt<2>::NODES);
	const uint32_t oNX = nx ? 48u : 0u, oFX = 48u - oNX, oNY = ny ? 64u : 16u, oFY = 80u - oNY, oNZ = nz ? 80u : 32u, oFZ = 112u - oNZ;
	#pragma unroll
	for (int g = 0; g < RL_LEAFLIST_RECORDS; ++g) {
		key[4 * g] = key[4 * g + 1] = key[4 * g + 2] = key[4 * g + 3] = 0xffffffffu;
		if (g < S.numLeafRecords) {   // the same for every lane
			c.nodes += 2;             // 64-byte records fetched
			RL_WSTEP(4);
			const char* rec = recs + g * (RL_LDS_NSTRIDE * 16);
			const float4 nX = *(const float4*)(rec + oNX), fX = *(const float4*)(rec + oFX);
			const float4 nY = *(const float4*)(rec + oNY), fY = *(const float4*)(rec + oFY);
			const float4 nZ = *(const float4*)(rec + oNZ), fZ = *(const float4*)(rec + oFZ);
			// (This form -- 34 issue cycles per box for 46 by the cost table of tools/valu_calib.hip -- ran SLOWER twice in the first half of round 3, 14.80 ms for 14.31, while
			// the kernel still parked its arguments in VGPR lanes; with those reloads gone (RL_ARGS) it is 13.25 ms for 13.52.)
			// One box: six fma, max + max3, min3, and the key.  The exit needs no clamp to FLT_MAX (an axis without a constraint gives +inf or NaN, which min3 skips;
			// "NaN * widen < tn" is false: the box counts as met), and the entry no clamp to 0: this kernel only runs with rayTMin >= 0 (rl_runtime.inl picks the
			// tree walk otherwise), so tn >= tMin >= 0 is a sortable key as it is.  RL_LL_SMEAR: "culled" as the sign of fma(exit, widen, -entry) smeared over the key.
			#ifndef RL_LL_SMEAR
			#define RL_LL_SMEAR 1
			#endif
			// (The smeared form works on the NEGATED entry distance, ntn = min(-tMin, -planes): the sign test is then fma(tf, widen, ntn) with the constant as the
			//  instruction's literal -- v_fmamk, 2 issue cycles; with "- tn" the compiler needs the three-operand encoding, which takes no literal, parks the
			//  constant in an SGPR and pays the 4 cycles of an SGPR operand -- and the key drops ntn's sign bit with the mask it applies anyway.)
			#define RL_LSLAB(k, slot) { \
				float tn = tMin, tf; \
				if (RL_LL_SMEAR) { \
					float ntn = -tMin; \
					ntn = fminf(ntn, -__builtin_fmaf(nX.k, invb.x, cn.x)); tf = __builtin_fmaf(fX.k, invb.x, cf.x); \
					ntn = fminf(ntn, -__builtin_fmaf(nY.k, invb.y, cn.y)); tf = fminf(tf, __builtin_fmaf(fY.k, invb.y, cf.y)); \
					ntn = fminf(ntn, -__builtin_fmaf(nZ.k, invb.z, cn.z)); tf = fminf(tf, __builtin_fmaf(fZ.k, invb.z, cf.z)); \
					key[slot] = ((__float_as_uint(ntn) & 0x7fffffe0u) | (uint32_t)(slot)) | (uint32_t)((int32_t)__float_as_uint(__builtin_fmaf(tf, RL_BOX_WIDEN, ntn)) >> 31); \
				} else { \
					tn = fmaxf(tn, __builtin_fmaf(nX.k, invb.x, cn.x)); tf = __builtin_fmaf(fX.k, invb.x, cf.x); \
					tn = fmaxf(tn, __builtin_fmaf(nY.k, invb.y, cn.y)); tf = fminf(tf, __builtin_fmaf(fY.k, invb.y, cf.y)); \
					tn = fmaxf(tn, __builtin_fmaf(nZ.k, invb.z, cn.z)); tf = fminf(tf, __builtin_fmaf(fZ.k, invb.z, cf.z)); \
					if (!(tf * RL_BOX_WIDEN < tn)) key[slot] = (__float_as_uint(tn) & ~31u) | (uint32_t)(slot); } }
			RL_LSLAB(x, 4 * g) RL_LSLAB(y, 4 * g + 1) RL_LSLAB(z, 4 * g + 2) RL_LSLAB(w, 4 * g + 3)
			#undef RL_LSLAB
		}
	}
	uint32_t from = 0u;   // keys below this one are done (keys are distinct: the slot is part of the key)
	// (One triangle per turn of ONE loop -- a lane picks its next leaf while its neighbours test their next triangle -- was measured too: 9.8
	// triangle steps per wave and bounce instead of 12 on 16 leaves, but 19.81 ms against 19.42: the pick costs more per turn than it saves.  Again on the
	// final kernel of round 3: 13.33 ms against 12.50.)
#ifdef RL_WATCHDOG
	int guardSel = 0;
#endif
	// the smallest key >= from, as the smallest (key - from) in unsigned arithmetic: an unused key (0xffffffff) lands on 0xffffffff - from and
	// a key below `from` (a leaf already visited) wraps around to more than that -- so "nothing left" is "the smallest is not below
	// 0xffffffff - from".  (Comparing the re-based minimum with 0xffffffff instead is wrong exactly when all 24 slots are candidates and
	// all have been visited: the minimum is then a wrapped one, never equals 0xffffffff, and the loop does not end.  tools/gpu_fuzz.py found it.)
	// The first pick (from == 0) needs no subtractions; the next one is made at the end of the loop's body.
	uint32_t m = 0xffffffffu;
	#pragma unroll
	for (int j = 0; j < 4 * RL_LEAFLIST_RECORDS; ++j) m = min(m, key[j]);
	for (;;) {
#ifdef RL_WATCHDOG
		if (++guardSel > 200) { printf("leaf-list pick stuck: lane %u from %u tMin %g best %g keys %u %u %u %u\n", threadIdx.x, from, tMin, best.t, key[0], key[1], key[2], key[3]); break; }
#endif
		if (m >= 0xffffffffu - from) break;
		m += from;
		// the nearest leaf left starts behind the hit (the slab test's own cut: tf * widen < tn; the keys are entry distances, rayTMin >= 0 here)
		if (best.t * RL_BOX_WIDEN < __uint_as_float(m & ~31u)) break;
		from = m + 1u;
		RL_WSTEP(6);
		const uint32_t j = m & 31u;
		const int ref = ((const int*)(sm + LdsAt<2>::NODES + (j >> 2) * RL_LDS_NSTRIDE + 6))[j & 3u];
		const uint32_t code = (uint32_t)~ref;
		const int first = (int)(code >> 6);
		const int count = (int)(code & 7u) + 1;
		const bool alpha = (code & 8u) != 0;
#if defined(RL_DIAG_STAMPS) && RL_DIAG_STAMPS >= 2
		// diagnostic build: what regrouping the (ray, triangle) pairs of this round across the wave could save at best.  The lanes that visit a leaf in this
		// round test `count` triangles each; dealt evenly to 64 lanes the round's pairs would take ceil(pairs / 64) wave steps instead of max(count) -- and no
		// fewer than one, because a ray's next leaf depends on what this one yields (the nearest-first cut).  Summed in slot 7 next to the steps taken (slot 5).
		{
			uint32_t pairs = 0;
			for (int cc = 1; cc <= 8; ++cc) pairs += (uint32_t)cc * (uint32_t)__popcll(Ballot(count == cc));
			const unsigned long long em_ = Ballot(true);
			if (c.diag && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)em_) - 1u) atomicAdd(&c.diag[CNT_COUNT + 7], (unsigned long long)((pairs + 63u) / 64u));
		}
#endif
		for (int i = 0; i < count; ++i) {
			const float4* tr = sm + LdsAt<2>::ISECT + (first + i) * 6;
			const float4 q0 = tr[0], q1 = tr[1], q2 = tr[2], q3 = tr[3];
			struct { V3 v0, n, u, v; float uv, uu, vv, denom, rden; } T;
			T.v0 = v3(q0.x, q0.y, q0.z); T.n = v3(q0.w, q1.x, q1.y); T.u = v3(q1.z, q1.w, q2.x); T.v = v3(q2.y, q2.z, q2.w);
			T.uv = q3.x; T.uu = q3.y; T.vv = q3.z; T.denom = q3.w; T.rden = tr[5].z;
			c.tris++;
			RL_WSTEP(5);
			const float t = dot((T.v0 - o), T.n) / dot(d, T.n);
			if (!(t >= tMin && t <= FLT_MAX && (t < best.t || (t == best.t && first + i < best.tri)))) continue;
			const V3 p = o + t * d;
			const V3 w = p - T.v0;
			const float wv = dot(w, T.v), wu = dot(w, T.u);
			float pa, pb;
			if (Barycentric(S.fastBary != 0, T.uv * wv - T.vv * wu, T.uv * wu - T.uu * wv, T.denom, T.rden, pa, pb) && OwnBoxPassMnMx(tr, o, v3(rtm::rcp1_(d.x), rtm::rcp1_(d.y), rtm::rcp1_(d.z)), tMin, t)) {
				if (alpha && !AlphaTestCandidate(S, first + i, pa, pb, c)) continue;
				best.t = t; best.a = pa; best.b = pb; best.tri = first + i;
				if (ANYHIT) return true;
			}
		}
		m = 0xffffffffu;
		#pragma unroll
		for (int j = 0; j < 4 * RL_LEAFLIST_RECORDS; ++j) m = min(m, key[j] - from);
	}
	return best.tri >= 0;
}

// The same closest-hit search on the BVH4 (DNode4): four slab tests per step, hit children ordered by entry distance.
// FULL: float boxes (S.nodes4f), else the grid nodes (S.nodes4)
template <int STACK, bool ANYHIT, bool PRIMS, bool FULL, int LDS = 0>
__device__ __forceinline__ bool Traverse4(const DSceneView& S, V3 o, V3 d, float rayTime, float tMin, HitRec& best, int* stk, Counters& c, const float4* sm = nullptr)
{
	if constexpr (LDS == 2) return TraverseLeafList<ANYHIT>(S, o, d, tMin, best, c, sm);
	c.rays++;
	V3 invb = v3(rtm::rcp1_(d.x), rtm::rcp1_(d.y), rtm::rcp1_(d.z));   // for the box tests (the candidate rule divides again: exact, and rare)
	if (!FULL) invb = ClampInv(invb);
	const bool nx = invb.x < 0.0f, ny = invb.y < 0.0f, nz = invb.z < 0.0f;
	best.t = INFINITY; best.tri = -1; best.a = 0.0f; best.b = 0.0f;
	int sp = 0, cur = 0;
	const int DONE = 0x7fffffff;
	for (;;) {
		while (cur >= 0 && cur != DONE) {
			c.nodes += FULL ? 2 : 1;   // 64-byte records fetched
			RL_WSTEP(4);
			const float tmx = fminf(best.t, FLT_MAX);
			float t0, t1, t2, t3; int r0, r1, r2, r3;
			if (FULL) { RL_WIDE_STEP_F((LDS ? sm + RL_LDS_NODES + cur * RL_LDS_NSTRIDE : (const float4*)(S.nodes4f + cur)), o, invb, nx, ny, nz, tMin, tmx, RL_BOX_WIDEN, a0, a1, a2, a3, ch) t0 = a0; t1 = a1; t2 = a2; t3 = a3; r0 = ch.x; r1 = ch.y; r2 = ch.z; r3 = ch.w; }
			else { RL_WIDE_STEP_Q(S, cur, o, invb, nx, ny, nz, tMin, tmx, RL_BOX_WIDEN, a0, a1, a2, a3, ch) t0 = a0; t1 = a1; t2 = a2; t3 = a3; r0 = ch.x; r1 = ch.y; r2 = ch.z; r3 = ch.w; }
			if (r0 == DNODE_EMPTY) t0 = INFINITY;
			if (r1 == DNODE_EMPTY) t1 = INFINITY;
			if (r2 == DNODE_EMPTY) t2 = INFINITY;
			if (r3 == DNODE_EMPTY) t3 = INFINITY;
			#define RL_CSWAPB(ta, ra, tb, rb) { const bool sw = tb < ta; const float tt = sw ? tb : ta; tb = sw ? ta : tb; ta = tt; const int rr = sw ? rb : ra; rb = sw ? ra : rb; ra = rr; }
			RL_CSWAPB(t0, r0, t1, r1) RL_CSWAPB(t2, r2, t3, r3) RL_CSWAPB(t0, r0, t2, r2) RL_CSWAPB(t1, r1, t3, r3) RL_CSWAPB(t1, r1, t2, r2)
			#undef RL_CSWAPB
			if (!(t0 < INFINITY)) { if (sp == 0) cur = DONE; else { --sp; cur = stk[sp * RL_BLOCK]; } continue; }
			if (t3 < INFINITY && sp < STACK) { stk[sp * RL_BLOCK] = r3; ++sp; }
			if (t2 < INFINITY && sp < STACK) { stk[sp * RL_BLOCK] = r2; ++sp; }
			if (t1 < INFINITY && sp < STACK) { stk[sp * RL_BLOCK] = r1; ++sp; }
			cur = r0;
		}
		if (cur == DONE) break;
		{
			const uint32_t code = (uint32_t)~cur;
			const int first = (int)(code >> 6);
			const int count = (int)(code & 7u) + 1;
			const bool alpha = (code & 8u) != 0;
			RL_WSTEP(6);
			for (int i = 0; i < count; ++i) {
				const Tri T = LDS ? TriFrom(sm + RL_LDS_ISECT + (first + i) * RL_LDS_TSTRIDE) : LoadTri(S, first + i);
				c.tris++;
				RL_WSTEP(5);
				const float t = dot((T.v0 - o), T.n) / dot(d, T.n);
				if (!(t >= tMin && t <= FLT_MAX && (t < best.t || (t == best.t && first + i < best.tri)))) continue;
				const V3 p = o + t * d;
				const V3 w = p - T.v0;
				const float wv = dot(w, T.v), wu = dot(w, T.u);
				float pa, pb;
				if (Barycentric(S.fastBary != 0, T.uv * wv - T.vv * wu, T.uv * wu - T.uu * wv, T.denom, T.rden, pa, pb) && OwnBoxPass(T.v0, T.v1, T.v2, o, v3(rtm::rcp1_(d.x), rtm::rcp1_(d.y), rtm::rcp1_(d.z)), tMin, t)) {
					if (alpha && !AlphaTestCandidate(S, first + i, pa, pb, c)) continue;
					best.t = t; best.a = pa; best.b = pb; best.tri = first + i;
					if (ANYHIT) return true;
				}
			}
		}
		if (sp == 0) break;
		--sp;
		cur = stk[sp * RL_BLOCK];
	}
	(void)rayTime;
	return best.tri >= 0;
}

// ---------------------------------------------------------------------------
// Surface interaction (reference geom/hit.h:16-36)
struct Surf { float t; V3 p, n; float U, V; V3 tangent, bitangent; };

// HitResult for the winning primitive (reference geom/triangle.cc:43-47, geom/sphere.cc:19-41, geom/cube.cc:24-38)
// + the tangent frame (geom/hit.cc:6-18).  Returns the material index.
template <bool PRIMS, int LDS = 0>
__device__ __forceinline__ int BuildSurface(const DSceneView& S, V3 o, V3 d, const HitRec& h, Surf& s, bool basis, Counters& c, const float4* sm = nullptr)
{
	int material;
	s.t = h.t;
	s.p = o + h.t * d;
	const uint32_t kind = PRIMS ? (((uint32_t)h.tri) >> 28) : 0u;
	if (kind == 0u) {
		const Shade sh = LDS ? ShadeFrom(sm + LdsAt<LDS>::SHADE + h.tri * RL_LDS_TSTRIDE) : LoadShade(S, h.tri);
		c.shaded++;
		const float a = h.a, b = h.b;
		s.n = normalize((1 - a - b) * sh.n0 + a * sh.n1 + b * sh.n2);
		s.U = (1 - a - b) * sh.s0 + a * sh.s1 + b * sh.s2;
		s.V = (1 - a - b) * sh.t0 + a * sh.t1 + b * sh.t2;
		material = sh.material;
	} else if (kind == 1u) {
		const float4* p = (const float4*)(S.spheres + (h.tri & 0x0fffffff));
		const float4 q = p[0];
		material = __float_as_int(p[1].x);
		c.shaded++;
		const V3 center = v3(q.x, q.y, q.z);
		s.n = (s.p - center) / q.w;
		const V3 op = s.p - center;
		s.U = rtm::atan_(op.y / op.x);
		s.V = rtm::acos_(op.z / q.w);
	} else {
		const float4* p = (const float4*)(S.cubes + (h.tri & 0x0fffffff));
		material = __float_as_int(p[1].w);
		c.shaded++;
		const int face = __float_as_int(h.a);
		s.n = (face == 0) ? v3(-1.0f, 0.0f, 0.0f) : (face == 1) ? v3(1.0f, 0.0f, 0.0f) : (face == 2) ? v3(0.0f, -1.0f, 0.0f)
		    : (face == 3) ? v3(0.0f, 1.0f, 0.0f) : (face == 4) ? v3(0.0f, 0.0f, -1.0f) : (face == 5) ? v3(0.0f, 0.0f, 1.0f) : v3s(0.0f);
		s.U = 0.0f; s.V = 0.0f;   // the reference leaves paramU / paramV unset for cubes
	}
	if (basis) {
		V3 T = (fabsf(s.n.x) > 0.9f) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
		V3 B = normalize(cross(T, s.n));
		T = normalize(cross(s.n, B));
		s.tangent = T; s.bitangent = B;
	}
	return material;
}
__device__ __forceinline__ V3 LocalToWorld(const Surf& s, V3 v)
{
	float wx = dot(v3(s.tangent.x, s.bitangent.x, s.n.x), v);
	float wy = dot(v3(s.tangent.y, s.bitangent.y, s.n.y), v);
	float wz = dot(v3(s.tangent.z, s.bitangent.z, s.n.z), v);
	return v3(wx, wy, wz);
}
__device__ __forceinline__ V3 WorldToLocal(const Surf& s, V3 v) { return v3(dot(v, s.tangent), dot(v, s.bitangent), dot(v, s.n)); }

// ---- microfacet BRDF pieces (reference render/brdf.h, render/material.cc:16-190) ----
__device__ __forceinline__ float Clampf(float val, float lo, float hi) { return fmaxf(lo, fminf(hi, val)); }

// (Inline since the end of round 3: as calls they measured better in round 2 (ErfInv / Erf inline 20.25 ms against 19.9), when the kernel had 400 SGPR reloads
//  to place around every call; with the arguments re-read per part of the loop, ErfInv + Erf + acosf inline are 12.26 ms against 12.45, 36.9 against 37.2 ms
//  on the 298 k frame.  powf stays a call: inline 12.65.)
#ifndef RL_ERFINV_ATTR
#define RL_ERFINV_ATTR __forceinline__
#endif
#ifndef RL_ERF_ATTR
#define RL_ERF_ATTR __forceinline__
#endif
__device__ RL_ERFINV_ATTR float ErfInv(float x)
{
	float w, p;
	x = Clampf(x, -.99999f, .99999f);
	w = -rtm::log_((1 - x) * (1 + x));
	if (w < 5) {
		w = w - 2.5f;
		p = 2.81022636e-08f;
		p = 3.43273939e-07f + p * w;
		p = -3.5233877e-06f + p * w;
		p = -4.39150654e-06f + p * w;
		p = 0.00021858087f + p * w;
		p = -0.00125372503f + p * w;
		p = -0.00417768164f + p * w;
		p = 0.246640727f + p * w;
		p = 1.50140941f + p * w;
	} else {
		w = rtm::sqrt_(w) - 3;
		p = -0.000200214257f;
		p = 0.000100950558f + p * w;
		p = 0.00134934322f + p * w;
		p = -0.00367342844f + p * w;
		p = 0.00573950773f + p * w;
		p = -0.0076224613f + p * w;
		p = 0.00943887047f + p * w;
		p = 1.00167406f + p * w;
		p = 2.83297682f + p * w;
	}
	return p * x;
}
__device__ RL_ERF_ATTR float Erf(float x)
{
	const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f;
	const float p = 0.3275911f;
	int sign = 1;
	if (x < 0) sign = -1;
	x = fabsf(x);
	float t = rtm::rcp1_(1 + p * x);
	float y = 1 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * rtm::exp_(-x * x);
	return sign * y;
}
__device__ __forceinline__ float SinThetaL(V3 w) { return rtm::sqrt_(fmaxf(0.0f, 1.0f - w.z * w.z)); }
__device__ __forceinline__ float CosPhi(V3 w) { float s = SinThetaL(w); return (s == 0) ? 1 : Clampf(w.x / s, -1, 1); }
__device__ __forceinline__ float SinPhi(V3 w) { float s = SinThetaL(w); return (s == 0) ? 0 : Clampf(w.y / s, -1, 1); }

// reference render/material.cc:83-165
__device__ void BeckmannSample11(float cosThetaI, float U1, float U2, float* slope_x, float* slope_y, Counters& cn)
{
	(void)cn;   // diagnostic builds count Newton iterations
	const float Pi = RL_PI;
	if ((double)cosThetaI > .9999) {
		float r = rtm::sqrt_(-rtm::log_(1.0f - U1));
		float sinPhi, cosPhi; rtm::sincos_(2 * Pi * U2, &sinPhi, &cosPhi);
		*slope_x = r * cosPhi;
		*slope_y = r * sinPhi;
		return;
	}
	float sinThetaI = rtm::sqrt_(fmaxf((float)0, (float)1 - cosThetaI * cosThetaI));
	float tanThetaI = sinThetaI / cosThetaI;
	float cotThetaI = rtm::rcp1_(tanThetaI);

	float a = -1, c = Erf(cotThetaI);
	float sample_x = fmaxf(U1, (float)1e-6f);

	float thetaI = rtm::acos_(cosThetaI);
	float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
	float b = c - (1 + c) * rtm::pow_(1 - sample_x, fit);

	const float SQRT_PI_INV = 1.f / sqrtf(Pi);
	float normalization = rtm::rcp1_(1 + c + SQRT_PI_INV * tanThetaI * rtm::exp_(-cotThetaI * cotThetaI));

	int it = 0;
	float invErf = 0.0f;
	bool converged = false;
	// The compiler unrolls this loop nine times whatever `#pragma nounroll` says (its trip count is a constant): 12 KB of code of which two or three copies ever
	// run.  Kept: with the limit hidden from the compiler (-DRL_NEWTON_ROLLED, round 5) the pool kernel is 10 KB shorter and every frame 0.6 - 1.1 % SLOWER
	// (298 k room from inside 89.3 against 88.4 ms, from outside 35.4 against 35.1, Cornell 12.39 against 12.30: profiles/r05_newton_rolled_ab.log).
	int newtonLimit = 10;
#ifdef RL_NEWTON_ROLLED
	asm volatile("" : "+s"(newtonLimit));
	#pragma nounroll
#endif
	while (++it < newtonLimit) {
		RL_WLSTEP(cn, 16, 17);
		if (!(b >= a && b <= c)) b = 0.5f * (a + c);
		invErf = ErfInv(b);
		float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * rtm::exp_(-invErf * invErf)) - sample_x;
		float derivative = normalization * (1 - invErf * tanThetaI);
		if (fabsf(value) < 1e-5f) { converged = true; break; }
		if (value > 0) c = b; else a = b;
		b -= value / derivative;
	}
	// the reference evaluates ErfInv(b) once more here (material.cc:163); after the break b is still the argument invErf was computed from,
	// so the value is in hand -- only a lane that used up its nine iterations has moved b since (a whole ErfInv per scattering event less)
	*slope_x = converged ? invErf : ErfInv(b);
	*slope_y = ErfInv(2.0f * fmaxf(U2, (float)1e-6f) - 1.0f);
}
// reference render/material.cc:166-190
__device__ V3 BeckmannSample(V3 wi, float alpha_x, float alpha_y, float U1, float U2, Counters& cn)
{
	V3 wiStretched = normalize(v3(alpha_x * wi.x, alpha_y * wi.y, wi.z));
	float slope_x, slope_y;
	BeckmannSample11(wiStretched.z, U1, U2, &slope_x, &slope_y, cn);
	float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
	slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
	slope_x = tmp;
	slope_x = alpha_x * slope_x;
	slope_y = alpha_y * slope_y;
	return normalize(v3(-slope_x, -slope_y, 1.f));
}
// reference render/brdf.h:39-58
__device__ __forceinline__ float DistributionBeckmann(V3 N, V3 H, float roughness)
{
	float cosH = dot(N, H);
	if (roughness == 0.0f) return 1.0f;
	if (H.z < 0.0f) cosH = -cosH;
	float cosH2 = cosH * cosH;
	float rr = roughness * roughness;
	float exp_x = (1.0f - cosH2) / (rr * cosH);
	float num = (cosH > 0.0f ? 1.0f : 0.0f) * rtm::exp_(-exp_x);
	float denom = RL_PI * rr * cosH2 * cosH2;
	return num / denom;
}
// reference render/brdf.h:74-93
__device__ __forceinline__ float GeometryBeckmann(V3 N, V3 H, V3 V, float roughness)
{
	float thetaV = rtm::acos_(dot(N, V));
	float tanThetaV = rtm::tan_(thetaV);
	float a = rtm::rcp1_(roughness * tanThetaV);
	float aa = a * a;
	if (dot(V, H) / dot(V, N) <= 0.0f) return 0.0f;
	if (a < 1.6f) {
		float num = 3.535f * a + 2.181f * aa;
		float denom = 1.0f + 2.276f * a + 2.577f * aa;
		return num / denom;
	}
	return 1.0f;
}

// material texture lookups (reference render/material.cc:297-303,378-395,406-415)
__device__ __forceinline__ V3 GetAlbedo(const DSceneView& S, const Mat& m, float U, float V, Counters& c)
{
	if (m.type == MAT_LAMBERTIAN || m.type == MAT_METAL) return m.albedo;
	if (m.type == MAT_MICROFACET) {
		V3 albedo = m.albedo;
		if (m.tex0 >= 0) { float4 px = TexSample(S, m.tex0, false, U, V, c); albedo = v3(px.x, px.y, px.z) * px.w; }   // tex0: the pow(2.2) copy made at upload
		return albedo;
	}
	return v3s(0.0f);
}
__device__ __forceinline__ float GetRoughness(const DSceneView& S, const Mat& m, float U, float V, Counters& c)
{
	float roughness = m.roughness;
	if (m.tex2 >= 0) roughness = TexSample(S, m.tex2, false, U, V, c).x;
	return roughness;
}
__device__ __forceinline__ V3 GetMicrosurfaceNormal(const DSceneView& S, const Mat& m, const Surf& s, Counters& c)
{
	if (m.type == MAT_MICROFACET && m.tex1 >= 0) {
		float4 px = TexSample(S, m.tex1, false, s.U, s.V, c);
		V3 N = v3(px.x, px.y, px.z);
		N = normalize(2.0f * N - 1.0f);
		return N;
	}
	return v3(0.0f, 0.0f, 1.0f);
}
__device__ __forceinline__ bool IsMirrorLike(const DSceneView& S, const Mat& m, float U, float V, Counters& c)
{
	if (m.type == MAT_DIELECTRIC || m.type == MAT_MIRROR) return true;
	if (m.type == MAT_MICROFACET) return GetRoughness(S, m, U, V, c) < 0.1f;
	return false;
}
// reference render/material.cc:342-350, material.h:67-69
__device__ __forceinline__ V3 Emitted(const DSceneView& S, const Mat& m, const Surf& s, Counters& c)
{
	if (m.type == MAT_DIFFUSE_LIGHT) return m.albedo;
	if (m.type == MAT_MICROFACET) {
		V3 emit = m.emissive;
		if (m.tex4 >= 0) { float4 px = TexSample(S, m.tex4, false, s.U, s.U, c); emit = v3s(px.z); }  // (U,U) and vec3(b): reference bugs kept
		return emit;
	}
	return v3s(0.0f);
}

// One scattering event.  Returns false when the material does not scatter.
// Outputs reflectance, new direction, pdf and ScatteringPdf (the value the
// reference recomputes at renderer.cc:144).
__device__ __forceinline__ bool Scatter(const DSceneView& S, const Mat& m, V3 inD, const Surf& s, Rng& g, Counters& c,
                                        V3& refl, V3& outD, float& pdf, float& sp)
{
	switch (m.type) {
		case MAT_DIFFUSE_LIGHT: return false;
		case MAT_LAMBERTIAN: {   // material.cc:195-219
			V3 N = s.n;
			V3 r = RandomInUnitSphere(g);
			if ((double)dot(r, N) < 0.0) r = -r;
			V3 Wi = normalize(r);
			outD = Wi;
			refl = m.albedo;
			pdf = absDot(N, Wi) / RL_PI;
			sp = fmaxf(0.0f, dot(s.n, Wi)) / RL_PI;
			return true;
		}
		case MAT_METAL: {        // material.cc:225-239
			V3 ud = normalize(inD);
			V3 reflected = reflect(ud, s.n);
			outD = reflected + m.fuzz * RandomInUnitSphere(g);
			refl = m.albedo;
			pdf = 1.0f;
			sp = 1.0f / RL_PI;
			return dot(outD, s.n) > 0.0f;
		}
		case MAT_MIRROR: {       // material.h:149-162
			refl = m.albedo;
			outD = reflect(inD, s.n);
			pdf = 1.0f;
			sp = 1.0f;
			return true;
		}
		case MAT_DIELECTRIC: {   // material.cc:244-285
			V3 outward_normal;
			V3 reflected = reflect(inD, s.n);
			float ni_over_nt;
			refl = m.transmission;
			V3 refracted = v3s(0.0f);
			float reflect_prob, cosine;
			if (dot(inD, s.n) > 0.0f) {
				outward_normal = -s.n;
				ni_over_nt = m.ior;
				cosine = m.ior * dot(inD, s.n) / length(inD);
			} else {
				outward_normal = s.n;
				ni_over_nt = rtm::rcp1_(m.ior);
				cosine = -dot(inD, s.n) / length(inD);
			}
			bool bRefract;
			{   // vec3.h:136-145
				V3 uv = normalize(inD);
				float dt = dot(uv, outward_normal);
				float D = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
				bRefract = D > 0.0f;
				if (bRefract) refracted = ni_over_nt * (uv - outward_normal * dt) - outward_normal * rtm::sqrt_(D);
			}
			if (bRefract) {
				float r0 = (1.0f - m.ior) / (1.0f + m.ior);
				r0 = r0 * r0;
				reflect_prob = r0 + (1.0f - r0) * rtm::pow_((1.0f - cosine), 5.0f);
			} else {
				reflect_prob = 1.0f;
			}
			outD = (Next(g) < reflect_prob) ? reflected : refracted;
			pdf = 1.0f;
			sp = 1.0f / RL_PI;
			return true;
		}
		default: {               // MicrofacetMaterial, material.cc:290-340,352-376,417-431
			RL_CSTAMP_BEGIN(c);
			RL_WLSTEP(c, 18, 19);
			V3 baseColor = GetAlbedo(S, m, s.U, s.V, c);
			float roughness = GetRoughness(S, m, s.U, s.V, c);
			float metallic = m.metallic;
			if (m.tex3 >= 0) metallic = TexSample(S, m.tex3, false, s.U, s.V, c).x;

			V3 N = GetMicrosurfaceNormal(S, m, s, c);
			V3 Wo = WorldToLocal(s, -inD);
			float u0 = Next(g);
			float u1 = Next(g);
			bool bFlip = Wo.z < 0.0f;
			RL_CSTAMP(c, 0);
			V3 Wh = BeckmannSample(bFlip ? -Wo : Wo, roughness, roughness, u0, u1, c);
			RL_CSTAMP(c, 1);
			if (bFlip) Wh = -Wh;
			V3 Wi = reflect(-Wo, Wh);
			float NdotWi = absDot(N, Wi);

			V3 F0 = v3s(0.04f);
			F0 = mix(F0, baseColor, metallic);
			V3 F = F0 + (1.0f - F0) * rtm::pow_(1.0f - absDot(Wh, Wo), 5.0f);
			float ggx2 = GeometryBeckmann(N, Wh, Wo, roughness);
			float ggx1 = GeometryBeckmann(N, Wh, Wi, roughness);
			float G = rtm::rcp1_(1.0f + ggx1 * ggx2);
			float NDF = DistributionBeckmann(N, Wh, roughness);

			V3 kS = F;
			V3 kD = 1.0f - kS;
			V3 diffuse = baseColor * (1.0f - metallic);
			V3 specular = (F * G * NDF) / (4.0f * NdotWi * absDot(N, Wo) + 0.001f);

			V3 WiW = LocalToWorld(s, Wi);
			outD = WiW;
			refl = (kD * diffuse + kS * specular) * NdotWi;

			// ScatteringPdf(hit, -inD, WiW), material.cc:352-376
			V3 wo = WorldToLocal(s, -inD);
			V3 wi = WorldToLocal(s, WiW);
			V3 wh = normalize(wo + wi);
			if (wh.z < 0.0f) wh.z = -wh.z;
			float D = DistributionBeckmann(N, wh, roughness);
			sp = D * absDot(wh, N);
			pdf = sp / (4.0f * dot(Wo, Wh));
			RL_CSTAMP(c, 2);
			return true;
		}
	}
}

// ---------------------------------------------------------------------------
// Camera::GetCameraRay (reference render/camera.h:44-53)
__device__ __forceinline__ void CameraRay(const DCamera& k, float s, float t, Rng& g, V3& o, V3& d, float& rayTime)
{
	V3 rd;
	if (k.lensRadius == 0.0f) {
		// Pinhole: lensRadius * RandomInUnitDisk() is a vector of zeros.  Only their SIGNS can still matter (a +-0 offset decides
		// the sign of an exactly-zero direction component), and those follow from the signs of cos/sin of the lens angle, which
		// do not need the polynomials.  The two draws are consumed as always (reference core/random.cc:42-50).
		(void)Next(g);
		const float u2 = Next(g);
		const float theta = 2.0f * 3.14159265358979323846f * u2;
		bool sn, cn; rtm::sincos_signs_(theta, &sn, &cn);
		rd = k.lensRadius * v3(cn ? -1.0f : 1.0f, sn ? -1.0f : 1.0f, 0.0f);
	} else {
		rd = k.lensRadius * RandomInUnitDisk(g);
	}
	V3 cu = ld3(k.u), cv = ld3(k.v);
	V3 offset = (cu * rd.x) + (cv * rd.y);
	float captureTime = k.beginTime + k.timePeriod * Next(g);
	rayTime = captureTime;   // ray.t: consumed by the moving Cube primitive (geom/cube.cc:5), inherited by scattered rays
	V3 origin = ld3(k.origin);
	o = origin + offset;
	d = normalize(ld3(k.top_left) + s * ld3(k.horizontal) + (1.0f - t) * ld3(k.vertical) - origin - offset);
}

// GenerateCell's pixel -> [0, 1) coordinates, reference render/renderer.cc:232-239:  u = x / W, v = y / H, each plus (Next() - 0.5) * 2 / W resp. H from the second
// sample on: four divisions by two constants of the launch, 144 issue cycles per camera ray.  With P.invWidth = RN(1 / W) they are rtm::div_by_'s 6 each, and its
// conditions hold without a guard: W, H are integers in [1, 2^32] as floats, the numerators are +0 or integers below 2^32 or multiples of 2^-23 in (-1, 1)
// (Next() is a multiple of 2^-24; x - 0.5 == 0 is +0) -- every quotient is +0 or at least 2^-55 in magnitude.
__device__ __forceinline__ void PixelUV(const DRenderParams& P, uint32_t x, uint32_t y, uint32_t sampleIndex, Rng& g, float& u, float& v)
{
	const float imageWidth = (float)P.width, imageHeight = (float)P.height;
	u = rtm::div_by_((float)x, imageWidth, P.invWidth);
	v = rtm::div_by_((float)y, imageHeight, P.invHeight);
	if (sampleIndex != 0) {
		u += rtm::div_by_((Next(g) - 0.5f) * 2.0f, imageWidth, P.invWidth);
		v += rtm::div_by_((Next(g) - 0.5f) * 2.0f, imageHeight, P.invHeight);
	}
}

struct SkyRot { float m0[3], m1[3], m2[3]; };   // Rotator(yaw 90).rotate rows, computed on the host (renderer.cc:166-168)

// Miss shader: sky panorama + sun (reference render/renderer.cc:155-199)
template <int STACK, bool PRIMS, bool FULL, int LDS = 0>
__device__ __forceinline__ V3 MissShader(const DSceneView& S, const SkyRot& R, V3 o, V3 d, float rayTime, float rayTMin, int* stk, Counters& c, const float4* sm = nullptr)
{
	V3 missResult = v3s(0.0f);
	if (S.sky) {
		V3 dir = normalize(d);
		V3 D = v3(dot(ld3(R.m0), dir), dot(ld3(R.m1), dir), dot(ld3(R.m2), dir));
		float u = rtm::atan2_(D.z, D.x), v = rtm::asin_(D.y);
		u *= 0.1591f; v *= 0.3183f;
		u += 0.5f; v += 0.5f;
		int x = (int)(u * (float)(uint32_t)(S.skyWidth - 1));
		int y = (int)(v * (float)(uint32_t)(S.skyHeight - 1));
		float4 px = ((const float4*)S.sky)[(uint32_t)(y * S.skyWidth + x)];
		c.texels++;
		missResult = missResult + v3(px.x, px.y, px.z);
	}
	if (S.hasSun) {
		HitRec tmp;
		bool occluded;
		if constexpr (LDS != 0) occluded = Traverse4<STACK, true, PRIMS, FULL, LDS>(S, o, -ld3(S.sunDirection), rayTime, rayTMin, tmp, stk, c, sm);
		else occluded = (!PRIMS && (FULL ? (const void*)S.nodes4f : (const void*)S.nodes4)) ? Traverse4<STACK, true, PRIMS, FULL, LDS>(S, o, -ld3(S.sunDirection), rayTime, rayTMin, tmp, stk, c, sm)
		                                           : Traverse<STACK, true, PRIMS>(S, o, -ld3(S.sunDirection), rayTime, rayTMin, tmp, stk, c);
		if (!occluded) missResult = missResult + ld3(S.sunIlluminance);
	}
	return missResult;
}

// one path's radiance in the sample buffer: 12 bytes (the buffer is written once and read once per sample: a fourth float would be a quarter more of both)
struct SampleRGB { float x, y, z; };
__device__ __forceinline__ SampleRGB make_sample(float x, float y, float z) { SampleRGB s; s.x = x; s.y = y; s.z = z; return s; }

// job -> (local cell, sample, pixel in cell) -> image coordinates
struct JobPixel { uint32_t x, y, slot, sample; bool valid; };
__device__ __forceinline__ JobPixel DecodeJob(const DRenderParams& P, uint32_t job)
{
	JobPixel j;
	const uint32_t p = job & 63u;
	const uint32_t rest = job >> 6;
	// n / d with d fixed per launch: q = mulhi(n, floor(2^32 / d)) is at most a few short; correct it
	uint32_t cellLocal = __umulhi(rest, P.magicSamples);
	uint32_t sLocal = rest - cellLocal * P.sampleCount;
	while (sLocal >= P.sampleCount) { sLocal -= P.sampleCount; ++cellLocal; }
#ifdef RL_EXP_STRIPS
	{   // experiment (whole frames whose cell columns divide by the heads only): a head's cells are a vertical STRIP of the frame, walked row by row
		const uint32_t cph = P.jobsPerHead / (P.sampleCount * 64u), hh = cellLocal / cph, ii = cellLocal % cph, sw = P.cellsX / P.numHeads;
		cellLocal = (ii / sw) * P.cellsX + hh * sw + ii % sw;
	}
#endif
	if (P.activeCells) cellLocal = P.activeCells[cellLocal];   // the job list holds the cells that can see the scene only (rl_device.h)
	const uint32_t cell = P.cellFirst + cellLocal * P.cellStride;
	uint32_t cy = __umulhi(cell, P.magicCellsX);
	uint32_t cx = cell - cy * P.cellsX;
	while (cx >= P.cellsX) { cx -= P.cellsX; ++cy; }
	j.x = cx * 8u + (p & 7u); j.y = cy * 8u + (p >> 3);
	j.slot = cellLocal * 64u + p;
	j.sample = sLocal;
	j.valid = (j.x < P.width) && (j.y < P.height);
	return j;
}

// The same for 64 consecutive jobs from a base that is a multiple of 64 (the leaf-list kernel's batches): one cell at one sample, the lane is the pixel.
// Everything but the pixel's coordinates is wave-uniform -- scalar arithmetic and, for the list of cells that can see the scene, a scalar load (the list
// is written before the launch: constant address space) -- where the per-lane form spends ~25 VALU instructions and a vector load whose s_waitcnt
// also waits for the wave's stores in flight.
__device__ __forceinline__ JobPixel DecodeJobBatch(const DRenderParams& P, uint32_t base, uint32_t lane)
{
	JobPixel j;
	const uint32_t rest = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) >> 6;
	uint32_t cellLocal = __umulhi(rest, P.magicSamples);
	uint32_t sLocal = rest - cellLocal * P.sampleCount;
	while (sLocal >= P.sampleCount) { sLocal -= P.sampleCount; ++cellLocal; }
	if (P.activeCells) cellLocal = *(const __attribute__((address_space(4))) uint32_t*)(P.activeCells + cellLocal);
	const uint32_t cell = P.cellFirst + cellLocal * P.cellStride;
	uint32_t cy = __umulhi(cell, P.magicCellsX);
	uint32_t cx = cell - cy * P.cellsX;
	while (cx >= P.cellsX) { cx -= P.cellsX; ++cy; }
	j.x = cx * 8u + (lane & 7u); j.y = cy * 8u + (lane >> 3);
	j.slot = cellLocal * 64u + lane;
	j.sample = sLocal;
	j.valid = (j.x < P.width) && (j.y < P.height);
	return j;
}

// ---------------------------------------------------------------------------
// The job list, sharded over the chip's XCDs.  An MI355X is 8 XCDs with a private, non-coherent 4 MiB L2 each; workgroups are dealt
// round-robin over them (MI355X_MICROARCH.md "Workgroup dispatch, XCD placement").  The job list (cell-major: all samples of local cell 0,
// then cell 1, ...) is cut into P.numHeads contiguous ranges of P.jobsPerHead jobs -- whole cells, so a range is a horizontal BAND of the
// frame (of this rank's cells) -- each behind a head word of its own, 128 bytes apart.  A wave draws from the head of the XCD it runs
// on (s_getreg HW_REG_XCC_ID): the camera rays an XCD's L2 sees come from one eighth of the image, i.e. they walk one region's part of
// the tree and its triangles, and eight words share the atomic traffic that one hot address took before.  When its own band is used up
// a wave steals from the band with the most jobs left (eight sc1 loads by eight lanes, a 3-step lane max), so the launch's end is
// worked on by everybody.  Heads count RELATIVE to their band's first job, so the host resets the whole queue with one memset.
// Results cannot depend on any of this: streams are keyed by (seed, pixel, sample).
// The reference's analogue is the single LIFO work queue of core/thread_pool.cc:93-112.
#define RL_HEAD_STRIDE 32u   /* uint32 words between two heads (128 B: one L2 line each) */
#define RL_MAX_HEADS 8u
__device__ __forceinline__ uint32_t XccId()
{
	return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;   // GETREG_IMMED(size - 1 = 3, offset 0, XCC_ID = 20): bits 3:0 of XCC_ID
}
struct JobSource { uint32_t cur, dry, left; };   // wave-uniform: the head this wave draws from; bit h: head h is known to be used up (it stays so); jobs that were left in `cur` after this wave's last draw
__device__ __forceinline__ JobSource JobSourceInit(const DRenderParams& P)
{
	JobSource js; js.cur = P.numHeads > 1u ? XccId() % P.numHeads : 0u; js.dry = 0u; js.left = 0xffffffffu;
	return js;
}
__device__ __forceinline__ uint32_t HeadLength(const DRenderParams& P, uint32_t h)
{
	const uint32_t first = h * P.jobsPerHead;   // (numHeads * jobsPerHead stays below 2^32: host side)
	return first < P.numJobs ? min(P.jobsPerHead, P.numJobs - first) : 0u;
}
// `want` (a multiple of 64) consecutive jobs for this wave: true with [base, end) set, false when every band is used up.  Wave-uniform.
__device__ __forceinline__ bool TakeJobs(const DRenderParams& P, unsigned int* __restrict__ heads, JobSource& js, uint32_t want, uint32_t lane, uint32_t& base, uint32_t& end)
{
	for (;;) {
		const uint32_t len = HeadLength(P, js.cur);
		// The end of a band in smaller pieces (P.guideShift > 0): a draw is at most 1 / 2^guideShift of what was left in the band after this
		// wave's previous draw there -- about half of "what is left / waves drawing from it" -- so that when the list runs dry a wave
		// holds a few batches, not a whole chunk of what may be the frame's dearest cells.  No extra read of the head: the size comes from
		// the wave's own last atomic (a stale upper bound: the band only shrinks).
		uint32_t ask = want;
		if (P.guideShift) ask = min(want, max(64u, (js.left >> P.guideShift) & ~63u));
		uint32_t rel = 0;
		if (lane == 0) rel = atomicAdd(&heads[js.cur * RL_HEAD_STRIDE], ask);
		rel = (uint32_t)__builtin_amdgcn_readfirstlane((int)rel);
		if (rel < len) {
			base = js.cur * P.jobsPerHead + rel; end = base + min(ask, len - rel);
			js.left = len - rel - min(ask, len - rel);
			return true;
		}
		js.dry |= 1u << js.cur;
		if (P.numHeads <= 1u) return false;
		// the fullest of the other bands.  A head only grows, so a stale value can only make a band look fuller than it is: the atomic
		// above then says so and the band is marked; "every band looks used up" is never wrong.
		uint32_t key = 0;
		if (lane < P.numHeads && !((js.dry >> lane) & 1u)) {
			const uint32_t nx = __hip_atomic_load(&heads[lane * RL_HEAD_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t ln = HeadLength(P, lane);
			key = nx < ln ? ((ln - nx) | lane) : 0u;   // jobs left (a multiple of 64) with the head's number in the low bits
		}
		key = max(key, (uint32_t)__shfl_xor((int)key, 1));
		key = max(key, (uint32_t)__shfl_xor((int)key, 2));
		key = max(key, (uint32_t)__shfl_xor((int)key, 4));
		key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
		if (key < 64u) return false;
		js.cur = key & 7u; js.left = key & ~63u;
	}
}

__device__ __forceinline__ void WaveLdsSync()
{
	// LDS operations of one wave are executed in issue order; this only stops the compiler from moving LDS
	// accesses of different lanes' data across the point.
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------
// The megakernel.  samples: [sampleCount][numLocalCells*64] SampleRGB.
// pathStack: [maxPathLength][stackStride] records of 2 float4 (refl.xyz, sp | pdf, E.xyz).
#ifndef RL_QUEUE_SHARED_CHUNK
#define RL_QUEUE_SHARED_CHUNK 1   /* leaf-list kernel: the workgroup's waves share one job chunk (see the refill) */
#endif
#ifndef RL_QUEUE_SPIN_LIMIT
// 0: a wave waits for the workgroup's chunk until the wave that is refilling it is done (microseconds: one global atomic).  N > 0: after N waits of 128
// cycles it takes one batch straight from the global counter instead (1: test build that always does; parity-tested).  The bounded form is not the
// default because its few instructions change the register allocation of the whole loop: 14.98 ms against 14.79 on the Cornell frame (same box, interleaved).
#define RL_QUEUE_SPIN_LIMIT 0
#endif
#ifndef RL_TRACE_MIN_WAVES
#define RL_TRACE_MIN_WAVES 4   /* 4 waves per SIMD = 4 workgroups per CU: caps the kernel at 128 VGPRs */
#endif
// PRIMS: the scene holds spheres / cubes (their leaf and shading code is compiled out of the triangle-only variant)
// FULL: the wide tree, if the launch carries one, has float boxes (S.nodes4f) -- small scenes; else grid nodes (S.nodes4)
// LDS (with FULL, triangle scenes within the RL_LDS_MAX* limits): the scene's records are copied to LDS at the start and read from there;
//     LDS == 2: a scene of <= 16 leaves, walked through its leaf list (TraverseLeafList) instead of its tree
// The megakernels' arguments are ~110 dwords of scalars (render parameters, camera, scene view, sky rotation, pointers).  The compiler loads them
// all at the kernel's entry and keeps them for its whole life -- in 102 SGPRs that the loops' own masks and counters need too: 50 of them went straight
// to VGPR lanes (v_writelane), and every later use was a v_readlane, four VALU issue cycles each, ~400 of them in k_trace's code (6 per record of the
// leaf list's box loop, 7 per pick, 9 per Newton iteration of the sampler ...), on a kernel that is bound by exactly that port.  RL_ARGS() reads them
// again from the kernel-argument segment where a part of the loop needs them (s_load through the scalar cache: no VALU slot, and three other waves
// to cover its latency): the offset goes through an empty asm statement, so that the loads can neither be hoisted out of the loop nor merged with
// the previous part's, and what they fetch dies with the block.  RL_KARG_RELOAD=0: the arguments as the compiler delivers them.
#ifndef RL_KARG_RELOAD
#define RL_KARG_RELOAD 1
#endif
struct KTraceArgs { DRenderParams P; DSceneView S; SkyRot R; SampleRGB* samples; float* pathStack; unsigned long long* counters; unsigned int* jobCounter; };
template <class T> __device__ __forceinline__ T KArg(uint32_t offset)
{
	uint32_t z = 0u;
	asm volatile("" : "+s"(z));
	T v;
	__builtin_memcpy(&v, (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offset + (z << 2), sizeof(T));   // (z << 2: the compiler must see a dword-aligned address to take the scalar path)
	return v;
}
#if RL_KARG_RELOAD
#define RL_ARGS() \
	const DRenderParams P = KArg<DRenderParams>((uint32_t)offsetof(KTraceArgs, P)); const DSceneView S = KArg<DSceneView>((uint32_t)offsetof(KTraceArgs, S)); \
	const SkyRot R = KArg<SkyRot>((uint32_t)offsetof(KTraceArgs, R)); SampleRGB* const samples = KArg<SampleRGB*>((uint32_t)offsetof(KTraceArgs, samples)); \
	float* const pathStack = KArg<float*>((uint32_t)offsetof(KTraceArgs, pathStack)); unsigned long long* const counters = KArg<unsigned long long*>((uint32_t)offsetof(KTraceArgs, counters)); \
	unsigned int* const jobCounter = KArg<unsigned int*>((uint32_t)offsetof(KTraceArgs, jobCounter)); \
	(void)P; (void)S; (void)R; (void)samples; (void)pathStack; (void)counters; (void)jobCounter
#else
#define RL_ARGS() \
	const DRenderParams& P = Pk; const DSceneView& S = Sk; const SkyRot& R = Rk; SampleRGB* const samples = samplesK; float* const pathStack = pathStackK; \
	unsigned long long* const counters = countersK; unsigned int* const jobCounter = jobCounterK; \
	(void)P; (void)S; (void)R; (void)samples; (void)pathStack; (void)counters; (void)jobCounter
#endif

template <int STACK, bool PRIMS, bool FULL, int LDS = 0>
__global__ void __launch_bounds__(RL_BLOCK, (STACK <= 32 ? RL_TRACE_MIN_WAVES : 2))
k_trace(const DRenderParams Pk, const DSceneView Sk, const SkyRot Rk, SampleRGB* __restrict__ samplesK,
        float* __restrict__ pathStackK, unsigned long long* __restrict__ countersK, unsigned int* __restrict__ jobCounterK)
{
	(void)Pk; (void)Sk; (void)Rk; (void)samplesK; (void)pathStackK; (void)countersK; (void)jobCounterK;
	RL_TEX_PROLOGUE(Sk);
	RL_MATH_PROLOGUE();
	__shared__ int s_stack[STACK * RL_BLOCK];
	__shared__ float4 s_scene[LDS ? LdsAt<LDS>::TOTAL : 1];
	const float4* sm = s_scene;
#if RL_QUEUE_SHARED_CHUNK
	// LDS == 2: the workgroup's four waves draw their batches of 64 jobs from ONE chunk (low word: next job, high word: end of the chunk)
	__shared__ unsigned long long s_jobs;
	__shared__ unsigned int s_lock, s_done;
	if (LDS == 2 && threadIdx.x == 0) { s_jobs = 0ull; s_lock = 0u; s_done = 0u; }   // empty: the first wave to ask draws the workgroup's first chunk from its XCD's head
#endif
	if (LDS) {
		RL_ARGS();
		const uint32_t nN = (uint32_t)(LDS == 2 ? S.numLeafRecords : S.numNodes4) * 8u, nT = (uint32_t)S.numTriangles * 4u, nM = (uint32_t)S.numMaterials * 5u;
		for (uint32_t i = threadIdx.x; i < 4u; i += RL_BLOCK) s_scene[RL_LDS_ROOT + i] = ((const float4*)S.nodes)[i];
		for (uint32_t i = threadIdx.x; i < nN; i += RL_BLOCK) s_scene[RL_LDS_NODES + (i >> 3) * RL_LDS_NSTRIDE + (i & 7u)] = ((const float4*)(LDS == 2 ? S.leafList : S.nodes4f))[i];
		if (LDS == 2) {
			for (uint32_t i = threadIdx.x; i < nT; i += RL_BLOCK) s_scene[LdsAt<LDS>::SHADE + i] = ((const float4*)S.shade)[i];
			for (uint32_t t = threadIdx.x; t < (uint32_t)S.numTriangles; t += RL_BLOCK) {   // the six-float4 record (LdsAt): edges and own box worked out here, once
				const Tri T = LoadTri(S, (int)t);
				const V3 mn = v3(fminf(fminf(T.v0.x, T.v1.x), T.v2.x), fminf(fminf(T.v0.y, T.v1.y), T.v2.y), fminf(fminf(T.v0.z, T.v1.z), T.v2.z));
				const V3 mx = v3(fmaxf(fmaxf(T.v0.x, T.v1.x), T.v2.x), fmaxf(fmaxf(T.v0.y, T.v1.y), T.v2.y), fmaxf(fmaxf(T.v0.z, T.v1.z), T.v2.z));
				float4* r = s_scene + LdsAt<LDS>::ISECT + t * 6u;
				r[0] = make_float4(T.v0.x, T.v0.y, T.v0.z, T.n.x); r[1] = make_float4(T.n.y, T.n.z, T.u.x, T.u.y); r[2] = make_float4(T.u.z, T.v.x, T.v.y, T.v.z);
				r[3] = make_float4(T.uv, T.uu, T.vv, T.denom); r[4] = make_float4(mn.x, mn.y, mn.z, mx.x); r[5] = make_float4(mx.y, mx.z, T.rden, 0.0f);
			}
		} else {
			for (uint32_t i = threadIdx.x; i < nT; i += RL_BLOCK) { const uint32_t at = (i >> 2) * RL_LDS_TSTRIDE + (i & 3u); s_scene[LdsAt<LDS>::ISECT + at] = ((const float4*)S.isect)[i]; s_scene[LdsAt<LDS>::SHADE + at] = ((const float4*)S.shade)[i]; }
		}
		for (uint32_t i = threadIdx.x; i < nM; i += RL_BLOCK) s_scene[LdsAt<LDS>::MATS + i] = ((const float4*)S.materials)[i];
		__syncthreads();
	}
	int* stk = s_stack + threadIdx.x;
	const uint32_t gtid = blockIdx.x * RL_BLOCK + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t numSlots;
	JobSource js;
	{ RL_ARGS(); numSlots = P.numLocalCells * 64u; js = JobSourceInit(P); }

	Counters c; c.rays = c.nodes = c.tris = c.shaded = c.texels = c.samples = c.trips = 0; RL_DIAG_BIND(c);
	Rng g; g.s.state = 0;
	V3 o = v3s(0.0f), d = v3s(0.0f);
	float rayTime = 0.0f;
	int depth = 0;
	uint32_t outIndex = 0;
	bool active = false;
	bool exhausted = false;

	// Wave-local job range: the wave takes P.jobChunk (64..1024) consecutive jobs from the global counter
	// with ONE atomic and deals them to its lanes itself.  (A returning atomic on one address
	// saturates near 88 dequeues/us chip-wide -- MI355X_MICROARCH.md "dequeue" -- and one atomic
	// per wave and bounce was exactly that rate: the kernel ran at the atomic's speed.)
	// (Round 2 gave every wave its first chunk without an atomic, because 4096 waves asking ONE counter at the same instant stood in line for ~45 us; with a
	// head per XCD the line is an eighth as long and the first chunk comes from the wave's own band like every other.)
	uint32_t chunkNext = 0, chunkEnd = 0;
	bool globalDone = false;
	uint32_t qCount = 0;   // LDS == 2: camera rays waiting in the wave's queue
	RL_TIMELINE(0);
#ifdef RL_DIAG_STAMPS
	// diagnostic build only: shader-clock time per phase (refill | traverse | shade | fold), summed per wave
	unsigned long long stampAcc[4] = { 0, 0, 0, 0 }, subAcc[4] = { 0, 0, 0, 0 }, laneAcc[4] = { 0, 0, 0, 0 }, laneT[4] = { 0, 0, 0, 0 };
	{ RL_ARGS(); c.diag = counters; }
	unsigned long long stampLast = __builtin_amdgcn_s_memtime();
	#define RL_SUBSTAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); subAcc[k] += now_ - subLast; subLast = now_; __builtin_amdgcn_sched_barrier(0); }
	unsigned long long subLast = 0;
	#define RL_STAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stampAcc[k] += now_ - stampLast; stampLast = now_; __builtin_amdgcn_sched_barrier(0); }
	// lane-weighted: clock x lanes that took part in the phase (k: 0 traverse, 1 shade a hit, 2 miss shader, 3 fold)
	#define RL_LANESTAMP(k, cond) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); laneAcc[k] += (now_ - laneLast) * (unsigned long long)__popcll(Ballot(cond)); laneT[k] += now_ - laneLast; __builtin_amdgcn_sched_barrier(0); }
	#define RL_LANEBEGIN() { __builtin_amdgcn_sched_barrier(0); laneLast = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
	unsigned long long laneLast = 0;
#else
	#define RL_STAMP(k)
	#define RL_SUBSTAMP(k)
	#define RL_LANESTAMP(k, cond)
	#define RL_LANEBEGIN()
#endif

#ifdef RL_WATCHDOG
	unsigned guardMain = 0;
#endif
	for (;;) {
#ifdef RL_WATCHDOG
		if (++guardMain > 200000u) { if (lane == 0) printf("k_trace main loop stuck: block %u wave %u active %llx exhausted %llx qCount %u globalDone %d chunk %u %u depth %d\n", blockIdx.x, threadIdx.x >> 6, (unsigned long long)Ballot(active), (unsigned long long)Ballot(exhausted), qCount, (int)globalDone, chunkNext, chunkEnd, depth); break; }
#endif
		// ---- refill idle lanes: wave64 ballot + prefix rank ----
		// Up to RL_REFILL_ROUNDS rounds: a fresh camera ray that misses both boxes of the root node can
		// only run the (sun-less) miss shader, so it is finished here and its lane takes another job at
		// once instead of occupying a lane slot through a whole bounce trip (in a 16:9 Cornell frame
		// more than half of the camera samples never touch the scene).
		if constexpr (LDS == 2) {
			RL_ARGS();
			// Leaf-list scenes need no traversal stack, and its LDS (1024 dwords per wave) is a QUEUE of camera rays instead: rays are
			// generated 64 at a time -- every lane takes a job, the same code for all of them -- the ones that cannot hit anything are finished
			// on the spot as in the rounds below, the others are written to the queue back to back (ballot + prefix rank), and the idle
			// lanes take theirs from its end.  The rounds below generate for the idle lanes only: a third of the wave in the first round, then a half
			// of that (in a 16:9 Cornell frame more than half of the camera samples miss the room), a quarter ... at the cost of a whole wave each time.
			enum { QCAP = 112, QFIELDS = 9 };   // 9 x 112 dwords <= 1024
			int* q = s_stack + (threadIdx.x >> 6) * (STACK * 64);
			const bool need = !active && !exhausted;
			const unsigned long long needMask = Ballot(need);
			const uint32_t n = (uint32_t)__popcll(needMask);
			while (n > 0 && qCount < n && qCount <= QCAP - 64 && !(globalDone && chunkNext >= chunkEnd)) {
#if RL_QUEUE_SHARED_CHUNK
				// The next batch of the workgroup's chunk: one LDS atomic.  The chunk is the granule of the GLOBAL job list (one global atomic per
				// P.jobChunk jobs, as before), the batch the granule of a wave's work: when the list runs dry a wave has at most its batch in
				// front of it, not a chunk -- the launch's tail shrinks from "one chunk per wave" to "a quarter of one".
				// Whoever finds the chunk used up takes the lock, asks the global counter and publishes the new chunk; the others wait for it.
				for (;;) {
					unsigned long long st = 0ull;
					if (lane == 0) st = atomicAdd(&s_jobs, 64ull);
					st = __shfl(st, 0);
					const uint32_t nx = (uint32_t)st, en = (uint32_t)(st >> 32);
					if (nx < en) { chunkNext = nx; chunkEnd = min(nx + 64u, en); break; }
					if (__atomic_load_n(&s_done, __ATOMIC_RELAXED) != 0u) { globalDone = true; chunkNext = chunkEnd = 0; break; }
					uint32_t won = 0;
					if (lane == 0) won = atomicCAS(&s_lock, 0u, 1u) == 0u ? 1u : 0u;
					won = __shfl(won, 0);
					if (won) {
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
						const unsigned long long cur = __atomic_load_n(&s_jobs, __ATOMIC_RELAXED);
						if ((uint32_t)cur >= (uint32_t)(cur >> 32) && __atomic_load_n(&s_done, __ATOMIC_RELAXED) == 0u) {   // still used up: nobody refilled it in between
							uint32_t base = 0, bend = 0;
							const bool got = TakeJobs(P, jobCounter, js, P.jobChunk, lane, base, bend);
							if (lane == 0) {
								if (!got) __atomic_store_n(&s_done, 1u, __ATOMIC_RELAXED);
								else __atomic_store_n(&s_jobs, (unsigned long long)base | ((unsigned long long)bend << 32), __ATOMIC_RELAXED);
							}
						}
						__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
						if (lane == 0) __atomic_store_n(&s_lock, 0u, __ATOMIC_RELAXED);
					} else {
						// wait for the wave that is asking the global counter (microseconds).  Not for ever: a wave that has waited ~1 ms stops relying
						// on its neighbours and takes one batch straight from the global counter -- always correct, the counter is the truth
#if RL_QUEUE_SPIN_LIMIT == 0
						while (__atomic_load_n(&s_lock, __ATOMIC_RELAXED) != 0u) __builtin_amdgcn_s_sleep(2);
						if (false) {
#else
						uint32_t spins = 0;
						while (__atomic_load_n(&s_lock, __ATOMIC_RELAXED) != 0u && ++spins < RL_QUEUE_SPIN_LIMIT) __builtin_amdgcn_s_sleep(2);
						if (spins >= RL_QUEUE_SPIN_LIMIT) {
#endif
							uint32_t base = 0, bend = 0;
							if (!TakeJobs(P, jobCounter, js, 64u, lane, base, bend)) { globalDone = true; chunkNext = chunkEnd = 0; }
							else { chunkNext = base; chunkEnd = bend; }
							break;
						}
					}
				}
				if (globalDone) { RL_TIMELINE(1); break; }
#else
				if (chunkNext >= chunkEnd) {
					uint32_t base = 0, bend = 0;
					if (!TakeJobs(P, jobCounter, js, P.jobChunk, lane, base, bend)) { globalDone = true; RL_TIMELINE(1); break; }
					chunkNext = base; chunkEnd = bend;
				}
#endif
				const uint32_t avail = chunkEnd - chunkNext;
				bool survive = false;
				V3 qo = v3s(0.0f), qd = v3s(0.0f); Rng qg; qg.s.state = 0; uint32_t qOut = 0;
				if (lane < avail) {
					const JobPixel j = DecodeJobBatch(P, chunkNext, lane);
					if (j.valid) {
						// GenerateCell body, reference render/renderer.cc:232-239
						const uint32_t sm_ = P.sampleBegin + j.sample;
						qg.s = raylib_rng_begin_mixed(P.seedMixed, j.y * P.width + j.x, sm_);
						float u, v;
						PixelUV(P, j.x, j.y, sm_, qg, u, v);
						float qTime;
						CameraRay(P.camera, u, v, qg, qo, qd, qTime);
						qOut = j.sample * numSlots + j.slot;
						c.samples++;
						survive = true;
						if (P.maxPathLength > 0 && RootMiss<LDS>(S, qo, qd, P.rayTMin, sm)) {
							const bool sunQuick = !S.hasSun || RootMiss<LDS>(S, qo, -ld3(S.sunDirection), P.rayTMin, sm);
							if (sunQuick) {
								c.rays++; c.nodes++;   // the closest-hit query this replaces fetches the root node and stops
								DSceneView Sq = S; Sq.hasSun = 0;
								V3 L = MissShader<STACK, PRIMS, FULL, LDS>(Sq, R, qo, qd, qTime, P.rayTMin, stk, c, sm);
								if (S.hasSun) { c.rays++; c.nodes++; L = L + ld3(S.sunIlluminance); }
								samples[qOut] = make_sample(L.x, L.y, L.z);
								survive = false;
							}
						}
					}
				}
				chunkNext += min(64u, avail);
				const unsigned long long sv = Ballot(survive);
				if (survive) {
					const uint32_t at = qCount + (uint32_t)__popcll(sv & ((1ull << lane) - 1ull));
					q[0 * QCAP + at] = __float_as_int(qo.x); q[1 * QCAP + at] = __float_as_int(qo.y); q[2 * QCAP + at] = __float_as_int(qo.z);
					q[3 * QCAP + at] = __float_as_int(qd.x); q[4 * QCAP + at] = __float_as_int(qd.y); q[5 * QCAP + at] = __float_as_int(qd.z);
					q[6 * QCAP + at] = (int)(uint32_t)qg.s.state; q[7 * QCAP + at] = (int)(uint32_t)(qg.s.state >> 32); q[8 * QCAP + at] = (int)qOut;
				}
				qCount += (uint32_t)__popcll(sv);
				WaveLdsSync();
			}
			if (need) {
				const uint32_t rank = (uint32_t)__popcll(needMask & ((1ull << lane) - 1ull));
				if (rank < qCount) {
					const uint32_t at = qCount - 1u - rank;
					o = v3(__int_as_float(q[0 * QCAP + at]), __int_as_float(q[1 * QCAP + at]), __int_as_float(q[2 * QCAP + at]));
					d = v3(__int_as_float(q[3 * QCAP + at]), __int_as_float(q[4 * QCAP + at]), __int_as_float(q[5 * QCAP + at]));
					g.s.state = (uint64_t)(uint32_t)q[6 * QCAP + at] | ((uint64_t)(uint32_t)q[7 * QCAP + at] << 32);
					outIndex = (uint32_t)q[8 * QCAP + at];
					rayTime = 0.0f;   // leaf-list scenes are triangle scenes: nothing moves, the ray's time is not read
					depth = 0;
					active = true;
				} else if (globalDone && chunkNext >= chunkEnd) exhausted = true;
			}
			qCount -= min(n, qCount);
			WaveLdsSync();
		} else
		for (int round = 0; round < RL_REFILL_ROUNDS; ++round) {
			RL_ARGS();
			const bool need = !active && !exhausted;
			const unsigned long long mask = Ballot(need);
			if (mask == 0ull) break;
			if (chunkNext >= chunkEnd && !globalDone) {
				uint32_t base = 0, bend = 0;
				if (!TakeJobs(P, jobCounter, js, P.jobChunk, lane, base, bend)) { globalDone = true; RL_TIMELINE(1); }
				else { chunkNext = base; chunkEnd = bend; }
			}
			const uint32_t avail = chunkEnd - chunkNext;
			if (need) {
				const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
				if (rank >= avail) {
					if (globalDone) exhausted = true;   // else: served in a later round / trip from the next chunk
				} else {
					const uint32_t job = chunkNext + rank;
					const JobPixel j = DecodeJob(P, job);
					if (j.valid) {
						// GenerateCell body, reference render/renderer.cc:232-239
						const uint32_t s = P.sampleBegin + j.sample;
						g.s = raylib_rng_begin_mixed(P.seedMixed, j.y * P.width + j.x, s);
						float u, v;
						PixelUV(P, j.x, j.y, s, g, u, v);
						CameraRay(P.camera, u, v, g, o, d, rayTime);
						depth = 0;
						outIndex = j.sample * numSlots + j.slot;
						active = true;
						c.samples++;
						if (P.maxPathLength > 0 && RootMiss<LDS>(S, o, d, P.rayTMin, sm)) {
							// The camera ray cannot hit anything.  Its miss shader (renderer.cc:155-199) is the sky lookup plus,
							// with a sun, one occlusion query from the ray origin; if that shadow ray misses the root too, the
							// whole sample is decided here.
							const bool sunQuick = !S.hasSun || RootMiss<LDS>(S, o, -ld3(S.sunDirection), P.rayTMin, sm);
							if (sunQuick) {
								c.rays++; c.nodes++;   // the closest-hit query this replaces fetches the root node and stops
								DSceneView Sq = S; Sq.hasSun = 0;
								V3 L = MissShader<STACK, PRIMS, FULL, LDS>(Sq, R, o, d, rayTime, P.rayTMin, stk, c, sm);
								if (S.hasSun) { c.rays++; c.nodes++; L = L + ld3(S.sunIlluminance); }
								samples[outIndex] = make_sample(L.x, L.y, L.z);
								active = false;
							}
						}
					}
				}
			}
			chunkNext += min((uint32_t)__popcll(mask), avail);
		}
		if (Ballot(active) == 0ull) {
			if (Ballot(!exhausted) == 0ull) break;
			continue;
		}

		// ---- one bounce for every active lane (TraceScene, reference render/renderer.cc:114-208) ----
		if (lane == 0) c.trips++;
		RL_STAMP(0);
		RL_LANEBEGIN();
		HitRec h; h.tri = -1;
		bool doTrace, hit = false;
		{
		RL_ARGS();
		doTrace = active && depth < P.maxPathLength;   // renderer.cc:120-123 otherwise
		// the 4-wide tree when the launch carries it (triangle scenes; half the steps: 24.6 -> 22.4 ms on the Cornell frame)
		if (doTrace) {
			if constexpr (LDS != 0) hit = Traverse4<STACK, false, PRIMS, FULL, LDS>(S, o, d, rayTime, P.rayTMin, h, stk, c, sm);   // an LDS-resident scene has its wide tree
			else hit = (!PRIMS && (FULL ? (const void*)S.nodes4f : (const void*)S.nodes4)) ? Traverse4<STACK, false, PRIMS, FULL, LDS>(S, o, d, rayTime, P.rayTMin, h, stk, c, sm) : Traverse<STACK, false, PRIMS>(S, o, d, rayTime, P.rayTMin, h, stk, c);
		}
		}
		RL_LANESTAMP(0, doTrace);
		RL_STAMP(1);
		if (active) {
			bool done = false, store = false;
			float4 rec0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rec1 = rec0;
			V3 L = v3s(0.0f);
			if (!doTrace) {
				done = true;
			} else if (hit) {
				RL_ARGS();
				Surf s;
				RL_LANEBEGIN();
#ifdef RL_DIAG_STAMPS
				subLast = __builtin_amdgcn_s_memtime();
#endif
				const int mi = BuildSurface<PRIMS, LDS>(S, o, d, h, s, true, c, sm);
				const Mat m = LDS ? MatFrom(sm + LdsAt<LDS>::MATS + mi * 5) : LoadMat(S, mi);
				RL_SUBSTAMP(0);
				V3 refl = v3s(0.0f), outD = v3s(0.0f);
				float pdf = 0.0f, sp = 0.0f;
				const bool scattered = Scatter(S, m, d, s, g, c, refl, outD, pdf, sp);
				RL_SUBSTAMP(1);
				const V3 E = Emitted(S, m, s, c);
				if (scattered && pdf > 0.0f) {
					if (depth + 1 >= P.maxPathLength) {
						// the next TraceScene returns 0 at once (renderer.cc:120-123): this vertex is the path's last, and its step of the
						// fold -- radiance = (0 + refl * 0 * sp / pdf) + E, the reference's expression -- is taken from the registers
						V3 radiance = v3s(0.0f);
						radiance = radiance + refl * L * sp / pdf;
						radiance = radiance + E;
						L = radiance;
						done = true;
					} else {
						// the vertex record (refl, sp | pdf, E) goes to the path stack BEHIND this trip's fold (below): a wave counts loads and
						// stores in one in-order counter, and a fold that waits for its loads behind this trip's stores waits for their write
						// acknowledgements too
						store = true;
						rec0 = make_float4(refl.x, refl.y, refl.z, sp);
						rec1 = make_float4(pdf, E.x, E.y, E.z);
						o = s.p; d = outD;
					}
				} else {
					L = v3s(0.0f) + E;                        // radiance(0) += Emitted, renderer.cc:137,151
					done = true;
				}
				RL_SUBSTAMP(2);
				RL_LANESTAMP(1, true);
			} else {
				RL_ARGS();
				RL_LANEBEGIN();
				L = MissShader<STACK, PRIMS, FULL, LDS>(S, R, o, d, rayTime, P.rayTMin, stk, c, sm);
				done = true;
				RL_LANESTAMP(2, true);
			}
			RL_STAMP(2);
			if (done) {
				RL_ARGS();
				RL_LANEBEGIN();
				// fold back to the camera: radiance = (0 + refl*Li*sp/pdf) + E at every vertex
#if RL_FOLD_PREFETCH > 0
				if (depth <= RL_FOLD_PREFETCH) {
					// all vertex records of the path are fetched before the dependent chain starts (one memory latency instead of one per vertex)
					float4 q0[RL_FOLD_PREFETCH], q1[RL_FOLD_PREFETCH];
					#pragma unroll
					for (int k = 0; k < RL_FOLD_PREFETCH; ++k) {
						const int kk = k < depth ? k : 0;
						const float4* st = (const float4*)pathStack + ((size_t)kk * P.stackStride + gtid) * 2u;
						q0[k] = st[0]; q1[k] = st[1];
					}
					#pragma unroll
					for (int k = RL_FOLD_PREFETCH - 1; k >= 0; --k) {
						if (k < depth) {
							const V3 refl = v3(q0[k].x, q0[k].y, q0[k].z);
							const float sp = q0[k].w, pdf = q1[k].x;
							const V3 E = v3(q1[k].y, q1[k].z, q1[k].w);
							V3 radiance = v3s(0.0f);
							radiance = radiance + refl * L * sp / pdf;
							radiance = radiance + E;
							L = radiance;
						}
					}
				} else
#endif
				for (int k = depth - 1; k >= 0; --k) {
					const float4* st = (const float4*)pathStack + ((size_t)k * P.stackStride + gtid) * 2u;
					const float4 r0 = st[0], r1 = st[1];
					const V3 refl = v3(r0.x, r0.y, r0.z);
					const float sp = r0.w, pdf = r1.x;
					const V3 E = v3(r1.y, r1.z, r1.w);
					V3 radiance = v3s(0.0f);
					radiance = radiance + refl * L * sp / pdf;
					radiance = radiance + E;
					L = radiance;
				}
				samples[outIndex] = make_sample(L.x, L.y, L.z);
				active = false;
				RL_LANESTAMP(3, true);
			}
			if (store) {
				RL_ARGS();
				// path vertex record: 32 contiguous bytes per lane, two 16-byte stores
				float4* st = (float4*)pathStack + ((size_t)depth * P.stackStride + gtid) * 2u;
				st[0] = rec0; st[1] = rec1;
				depth++;
			}
		}
		RL_STAMP(3);
	}

	RL_ARGS();
#ifdef RL_DIAG_STAMPS
	if (lane == 0) for (int k = 0; k < 4; ++k) { atomicAdd(&counters[CNT_COUNT + k], stampAcc[k]); atomicAdd(&counters[CNT_COUNT + 8 + k], subAcc[k]); atomicAdd(&counters[CNT_COUNT + 12 + k], c.tAcc[k]); atomicAdd(&counters[CNT_COUNT + 20 + k], laneAcc[k]); if (RL_DIAG_STAMPS < 2) atomicAdd(&counters[CNT_COUNT + 4 + k], laneT[k]); }
#endif
	RL_TIMELINE(2);
	// ---- counters: wave reduction, one atomic per wave and counter ----
	uint32_t vals[CNT_COUNT] = { c.rays, c.nodes, c.tris, c.shaded, c.texels, c.samples, c.trips };
	for (int k = 0; k < CNT_COUNT; ++k) {
		unsigned long long v = vals[k];
		for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
		if (lane == 0 && v) atomicAdd(&counters[k], v);
	}
}

// ---------------------------------------------------------------------------
// The pool megakernel.  Same job queue, same per-path arithmetic and the same outputs as k_trace, but a wave
// no longer runs "one ray per lane per trip".  Each wave owns a POOL of 64*K paths:
//   - the per-ray data the traversal needs (origin, direction, time) and gives back (t, primitive, barycentrics)
//     sit in LDS, one column per pool slot;
//   - the rest of a path (RNG state, output index, depth) stays in the registers of the slot's HOME lane
//     (slot = p*64 + lane), its vertex records in the global path stack.
// A trip is: refill the free slots (wave64 ballot + prefix ranks, compacted: up to 64 new camera rays are generated
// by the low lanes and dealt to the free slots through LDS), then ONE traversal phase over the whole pool, then K
// shading passes.  In the traversal phase a lane takes the next un-traced slot from the pool whenever it has
// finished its ray ("dynamic fetch"; the ray's home lane is irrelevant), so a wave's traversal time is the
// SUM of its rays' steps / 64 plus a tail, instead of the MAX over lanes per bounce.  The sun query of the miss
// shader (renderer.cc:192-197) goes through the pool like any other ray instead of being traced inline by the few
// lanes that missed.
enum { F_OX = 0, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_T, F_TRI, F_A, F_B, F_TIME, F_COUNT };   // F_TIME only exists in scenes with moving primitives (PRIMS)
#define Q_CLOSEST  (-1)   /* F_TRI before traversal: closest-hit query; after: missed everything */
#define Q_SHADOW   (-2)   /* before: occlusion query towards the sun (sky part parked in F_D*); after: not occluded */
#define Q_EMPTY    (-3)   /* no path in this slot */
#define Q_OCCLUDED (-4)   /* after a Q_SHADOW query: something is in the way */
#define Q_PENDING  (-5)   /* a lane is tracing this slot's closest-hit query (it may take more than one trip) */
#define Q_PENDING_SHADOW (-6)
#define Q_MISS     (-7)   /* result of a closest-hit query that hit nothing (distinct from Q_CLOSEST: a straggler may deliver it while the next phase is handing out slots) */
#define Q_CLEAR    (-8)   /* result of a sun query: nothing in the way */
#define RL_POOL_WIDEN RL_BOX_WIDEN
#ifndef RL_POOL_SHORT_LSTACK
#define RL_POOL_SHORT_LSTACK 18   /* LDS entries of the "short" 32-deep stack: 18 KiB + 20.5 KiB pool + 640 B of libm tables = 4 workgroups per CU */
#endif
#define RL_POOL_SHORT_MAXDEPTH 24 /* BVH depth up to which the short variant is used (deeper trees overflow too often: measured) */
#ifndef RL_POOL_MAXBLOCKS
#define RL_POOL_MAXBLOCKS 4   /* workgroups per CU the pool kernel is compiled for (register budget 512 / (4 * blocks) per lane) */
#endif
// Re-tuned in round 2 on the grid nodes (tools/gpu_variants.py, tools/gpu_scenes_time.py; 40 / 40 / 52 before): 298 k scene 51.9 ->
// 50.9 ms, colonnade 481 -> 471 ms, 2.36 M 156 -> 154 ms, 10.1 M 463 -> 460 ms.  (Cut 24: colonnade 458 but 2.36 M 159; cut 16: 459 / 163.)
#ifndef RL_POOL_CUT_EXH
#define RL_POOL_CUT_EXH 32   /* the same once the job queue is empty */
#endif
#ifndef RL_POOL_CUT
#define RL_POOL_CUT 32    /* with the pool handed out: shade once no more than this many lanes still traverse */
#endif
#ifndef RL_POOL_WNODE
#define RL_POOL_WNODE 4   /* relative cost of a node step and a primitive step in the vote */
#define RL_POOL_WLEAF 5
#endif
#ifndef RL_POOL_WNODE4
#define RL_POOL_WNODE4 4  /* the same for a BVH4 step */
#endif
#ifndef RL_POOL_WNODE8
#define RL_POOL_WNODE8 4  /* ... and for a step on the 8-wide tree */
#endif
#ifndef RL_POOL_BOTH8
#define RL_POOL_BOTH8 0
#endif
#ifndef RL_POOL_SHADE_MIN
// Hits wait in their pool slots until a shading round is worth running.  Until round 5 that meant a full wave of 64: the material code then always ran with every lane, and on
// average half a round's worth of finished hits -- a quarter of the pool's 128 slots -- sat parked instead of holding rays for the traversal phase, whose lanes run
// dry towards its end.  From 32 waiting hits on a round runs at once: 298 k room from inside 88.3 -> 85.5 ms, from outside 34.9 -> 33.8, colonnade 375.6 -> 363.6,
// 2.36 M 97.4 -> 94.5, 10.1 M 271.1 -> 267.2, textured room 102.0 -> 98.1 (thresholds 8 ... 48 are within 0.5 % of each other; profiles/r05_shade_min_ab.log).
#define RL_POOL_SHADE_MIN 32
#endif
#ifndef RL_POOL_WLEAF8
#define RL_POOL_WLEAF8 12   /* 298 k-triangle room from inside: 6 -> 365.7 ms, 9 -> 358.9, 12 -> 357.8, 16 -> 360.9 (the 4-wide tree: 381.4) */
#endif
#ifndef RL_POOL_WLEAF4
// Re-tuned at the end of round 3 (the leaf step is a third cheaper than it was -- two divisions gone, the own-box rule on v_max / v_min -- but above all the lanes at
// leaves are the ones about to FINISH: serving them first frees lanes for the next fetch).  298 k frame / colonnade / 2.36 M triangles at 4K, ms: 5 -> 36.85 / 373.7 /
// 114.2; 7 -> 36.0 / 372.9 / --; 8 -> 35.87 / 374.1 / 109.9; 10 -> 35.64 / 377.4 / 108.5; 12 -> 35.6 / -- / --; 16 -> 35.85 / 390.4 / 107.5.
#define RL_POOL_WLEAF4 8
#endif
#ifndef RL_POOL_KEEP
#define RL_POOL_KEEP 58   /* leave the traversal loop to fetch new rays when no more than this many lanes still traverse */
#endif


struct Trav {
	V3 o, d, inv; float rayTime; bool nx, ny, nz, anyhit; HitRec best; int cur, sp, leafI;
	// the 8-wide tree's walk (NodeStep8 / LeafStep8): the hit inner children of a node still to be visited, as ONE group -- gx the node's childBase, gy = their bits
	// in VISITING order (bit 24 + (slot XOR oct), highest first) | the node's alphaMask << 8 | its imask --; the triangles of its hit leaf children still to be tested:
	// tx the node's triBase, tz its leafMask, ty the bits of tz that are left; oct: bit 0 / 1 / 2 set when the ray travels towards +x / +y / +z
	uint32_t gx, gy, tx, ty, tz, oct;
	// ... and per axis all ones where the ray travels in the negative direction (NodeStep8 selects a node's near / far planes with them)
	uint32_t m8x, m8y, m8z;
};

// Single steps on the resumable state, for the vote-driven loop of k_trace_pool: a lane is either at an inner node
// (cur >= 0), at a leaf (cur < 0, leafI = next primitive of it), or finished (both return true then).
// LSTACK entries of the traversal stack live in LDS (stk), deeper ones in the lane's private overflow array (scratch):
// with a 19-entry LDS part a 32-deep stack fits 4 workgroups per CU; trees rarely need the overflow.
// (Round 3, measured and not kept: a 16-bit entry distance beside every stack entry, so that a pop drops the entries that start behind the best hit without
// fetching their node.  It drops next to nothing -- 6.8 node records per ray instead of 6.9 on the 298 k-triangle scene: the near-first walk with its
// shrinking t leaves little behind -- and the column costs LDS stack depth (12 entries instead of 18): 51.6 ms against 44.7.)
template <int LSTACK, int STACK>
__device__ __forceinline__ void StackPush(Trav& T, int* stk, int* ovf, int v)
{
	if (T.sp < LSTACK) { stk[T.sp * RL_BLOCK] = v; ++T.sp; }
	else if (LSTACK < STACK && T.sp < STACK) { ovf[T.sp - LSTACK] = v; ++T.sp; }
}
#ifndef RL_POP_SPLIT
#define RL_POP_SPLIT 0
#endif
template <int LSTACK, int STACK>
__device__ __forceinline__ bool PopOrFinish(Trav& T, int* stk, int* ovf)
{
	if (T.sp == 0) return true;
	--T.sp;
	// (The compiler sinks the two loads, one from scratch and one from LDS, into ONE flat_load through a selected pointer.  RL_POP_SPLIT keeps them apart --
	//  measured: the colonnade hall, whose rays live above the LDS part of the stack, 402 ms against 377: two divergent arms cost more than the flat load.)
#if RL_POP_SPLIT
	int v;
	if (LSTACK < STACK && T.sp >= LSTACK) { v = ovf[T.sp - LSTACK]; asm volatile("" : "+v"(v)); }
	else v = stk[T.sp * RL_BLOCK];
	T.cur = v;
#else
	T.cur = (LSTACK < STACK && T.sp >= LSTACK) ? ovf[T.sp - LSTACK] : stk[T.sp * RL_BLOCK];
#endif
	T.leafI = 0;
	return false;
}
// min(t, FLT_MAX) for a t that is never NaN (a hit distance, or +inf): one integer minimum on the bit patterns -- floats below FLT_MAX, negative ones
// included, are below 0x7f7fffff as signed integers too -- where fminf costs the compiler's canonicalising v_max t, t in front of the v_min
__device__ __forceinline__ float ClampToFltMax(float t) { return __int_as_float(min(__float_as_int(t), 0x7f7fffff)); }
template <int LSTACK, int STACK>
__device__ __forceinline__ bool NodeStep(const DSceneView& S, Trav& T, float tMin, int* stk, int* ovf, Counters& c)
{
	RL_WSTEP(4);
	const float4* np = (const float4*)(S.nodes + T.cur);
	const float4 q0 = np[0], q1 = np[1], q2 = np[2];
	const int4 k = ((const int4*)np)[3];
	c.nodes++;
	float tl, tr;
	const float tmx = ClampToFltMax(T.best.t);
	bool hl = Slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, T.o, T.inv, T.nx, T.ny, T.nz, tMin, tmx, tl, RL_POOL_WIDEN);
	bool hr = Slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, T.o, T.inv, T.nx, T.ny, T.nz, tMin, tmx, tr, RL_POOL_WIDEN);
	hl = hl && (k.x != DNODE_EMPTY);
	hr = hr && (k.y != DNODE_EMPTY);
	T.leafI = 0;
	if (hl && hr) {
		const bool leftFirst = tl <= tr;
		const int nearC = leftFirst ? k.x : k.y, farC = leftFirst ? k.y : k.x;
		StackPush<LSTACK, STACK>(T, stk, ovf, farC);
		T.cur = nearC;
		return false;
	}
	if (hl) { T.cur = k.x; return false; }
	if (hr) { T.cur = k.y; return false; }
	return PopOrFinish<LSTACK, STACK>(T, stk, ovf);
}
// One step on the BVH4 (DNode4, 128 B): four slab tests, the hit children ordered by entry distance (5-comparator network),
// the nearest followed, the others pushed far-to-near.  Counts as two 64-byte node records.
template <int LSTACK, int STACK>
__device__ __forceinline__ bool NodeStep4(const DSceneView& S, Trav& T, float tMin, int* stk, int* ovf, Counters& c)
{
	RL_WSTEP(4);
	c.nodes += RL_Q4 ? 1 : 2;   // 64-byte records fetched
	const float tmx = ClampToFltMax(T.best.t);
#if RL_Q4
	RL_WIDE_STEP_Q(S, T.cur, T.o, T.inv, T.nx, T.ny, T.nz, tMin, tmx, RL_POOL_WIDEN, t0, t1, t2, t3, ch)   // T.inv was clamped when the ray was fetched
#else
	RL_WIDE_STEP_F((const float4*)(S.nodes4f + T.cur), T.o, T.inv, T.nx, T.ny, T.nz, tMin, tmx, RL_POOL_WIDEN, t0, t1, t2, t3, ch)
#endif
	int r0 = ch.x, r1 = ch.y, r2 = ch.z, r3 = ch.w;
	if (r0 == DNODE_EMPTY) t0 = INFINITY;
	if (r1 == DNODE_EMPTY) t1 = INFINITY;
	if (r2 == DNODE_EMPTY) t2 = INFINITY;
	if (r3 == DNODE_EMPTY) t3 = INFINITY;
	#define RL_CSWAP(ta, ra, tb, rb) { const bool sw = tb < ta; const float tt = sw ? tb : ta; tb = sw ? ta : tb; ta = tt; const int rr = sw ? rb : ra; rb = sw ? ra : rb; ra = rr; }
	RL_CSWAP(t0, r0, t1, r1) RL_CSWAP(t2, r2, t3, r3) RL_CSWAP(t0, r0, t2, r2) RL_CSWAP(t1, r1, t3, r3) RL_CSWAP(t1, r1, t2, r2)
	#undef RL_CSWAP
	T.leafI = 0;
	if (!(t0 < INFINITY)) return PopOrFinish<LSTACK, STACK>(T, stk, ovf);
	if (t3 < INFINITY) StackPush<LSTACK, STACK>(T, stk, ovf, r3);
	if (t2 < INFINITY) StackPush<LSTACK, STACK>(T, stk, ovf, r2);
	if (t1 < INFINITY) StackPush<LSTACK, STACK>(T, stk, ovf, r1);
	T.cur = r0;
	return false;
}

template <int LSTACK, int STACK, bool PRIMS>
__device__ __forceinline__ bool LeafStep(const DSceneView& S, Trav& T, float tMin, int* stk, int* ovf, Counters& c)
{
	RL_WSTEP(5);
	const uint32_t code = (uint32_t)~T.cur;
	const int first = (int)(code >> 6);
	const int count = (int)(code & 7u) + 1;
	const bool alpha = (code & 8u) != 0;
	const uint32_t kind = (code >> 4) & 3u;
	const V3 o = T.o, d = T.d;
	c.tris++;
	if (!PRIMS || kind == 0u) {
		const int i = first + T.leafI;
		const Tri TT = LoadTri(S, i);
		// reference geom/triangle.cc:22-27
		const float t = dot((TT.v0 - o), TT.n) / dot(d, TT.n);
		if (t >= tMin && t <= FLT_MAX && (t < T.best.t || (t == T.best.t && i < T.best.tri))) {   // ties: the lower slot, as in Traverse()
			const V3 pp = o + t * d;
			const V3 w = pp - TT.v0;
			const float wv = dot(w, TT.v), wu = dot(w, TT.u);
			float pa, pb;
			if (Barycentric(S.fastBary != 0, TT.uv * wv - TT.vv * wu, TT.uv * wu - TT.uu * wv, TT.denom, TT.rden, pa, pb) && OwnBoxPass(TT.v0, TT.v1, TT.v2, o, v3(rtm::rcp1_(d.x), rtm::rcp1_(d.y), rtm::rcp1_(d.z)), tMin, t)) {
				if (!alpha || AlphaTestCandidate(S, i, pa, pb, c)) {
					T.best.t = t; T.best.a = pa; T.best.b = pb; T.best.tri = i;
					if (T.anyhit) return true;
				}
			}
		}
	} else {
		float2 r;
		if (kind == 1u) r = make_float2(SphereHit(S.spheres, first, o, d, tMin, T.best.t), 0.0f);
		else r = CubeHit(S.cubes, first, o, d, T.rayTime, tMin, T.best.t);
		if (r.x == r.x) {   // not NaN: a hit
			T.best.t = r.x; T.best.a = r.y; T.best.b = 0.0f; T.best.tri = (int)((kind << 28) | (uint32_t)first);
			if (T.anyhit) return true;
		}
	}
	if (++T.leafI < count) return false;
	return PopOrFinish<LSTACK, STACK>(T, stk, ovf);
}

// ---- the 8-wide tree (DNode8, rl_device.h) in the vote-driven loop --------------------------------------------------------------------------------
// A lane's state is (T.gx, T.gy): the group of hit inner children it is working through, (T.tx, T.ty, T.tz): the triangles of hit leaf children it still has
// to test, and a stack of groups (two words each: the LDS stack's entries pairwise, then the private overflow).  T.cur only says which party of the vote the
// lane belongs to: 0 at a node (a group with a child left), -1 at a leaf (a triangle left), TRAV-idle without a ray.  One node step = take the group's next
// child in visiting order, push the rest of the group (ONE entry however many children it holds), fetch the child (five 16-byte loads), test its eight boxes,
// and turn the hits into the next group and the next triangles -- no sort, no per-child pushes.  The triangles of a node's leaf children are tested before any
// of its inner children is entered (they are the geometry nearest to hand); the order of two candidates never decides a hit (candidate rule, tie rule).
// T.gy = the group's bits in VISITING order (bit 24 + (slot XOR oct), highest first) | the node's imask; T.ty = the hit leaf children (bits 0 - 7, slot order) |
// the next triangle of the lowest of them (bits 8 - 9) | the node's alphaMask << 16; T.tx / T.tz = the node's triBase / leafMask.
__device__ __forceinline__ void Push8(Trav& T, int* stk, int* ovf, const int G, const int GMAX)
{
	if (T.sp < G) { stk[(2 * T.sp) * RL_BLOCK] = (int)T.gx; stk[(2 * T.sp + 1) * RL_BLOCK] = (int)T.gy; ++T.sp; }
	else if (T.sp < GMAX) { ovf[2 * (T.sp - G)] = (int)T.gx; ovf[2 * (T.sp - G) + 1] = (int)T.gy; ++T.sp; }   // (GMAX = RL_POOL8_MAXLEVELS: the host selects this walk only for trees of at most that many levels, rl_runtime.inl SelectTraceKernel)
}
// what comes next for a lane whose triangles are done: the rest of its group, else the stack's top group, else nothing (true: the ray is finished)
__device__ __forceinline__ bool Next8(Trav& T, int* stk, int* ovf, const int G)
{
	if ((T.ty & 0xffu) != 0u) { T.cur = -1; return false; }
	if ((T.gy >> 24) != 0u) { T.cur = 0; return false; }
	if (T.sp == 0) return true;
	--T.sp;
	if (T.sp < G) { T.gx = (uint32_t)stk[(2 * T.sp) * RL_BLOCK]; T.gy = (uint32_t)stk[(2 * T.sp + 1) * RL_BLOCK]; }
	else { T.gx = (uint32_t)ovf[2 * (T.sp - G)]; T.gy = (uint32_t)ovf[2 * (T.sp - G) + 1]; }
	T.cur = 0;
	return false;
}
// a ray's constants for this walk: the octant (visiting order = slot XOR oct) and, per axis, all ones where the ray travels in the negative direction -- the
// near planes of a node are then (upper & m) | (lower & ~m): one v_bitop3_b32, 2 issue clocks, where a v_cndmask on a lane mask in SGPRs takes 4
__device__ __forceinline__ void RaySetup8(Trav& T)
{
	T.m8x = T.inv.x < 0.0f ? 0xffffffffu : 0u; T.m8y = T.inv.y < 0.0f ? 0xffffffffu : 0u; T.m8z = T.inv.z < 0.0f ? 0xffffffffu : 0u;
	T.oct = (T.inv.x < 0.0f ? 0u : 1u) | (T.inv.y < 0.0f ? 0u : 2u) | (T.inv.z < 0.0f ? 0u : 4u);
}
// One child: six planes, the NEGATED entry distance = min of the negated near distances (fma(q, -A, -(B - E)): the modifier is free), exit = min of the far ones,
// and "culled" (exit * widen < entry in real arithmetic) as the SIGN of fma(exit, widen, -entry) -- with the entry negated the widening is the instruction's literal
// (v_fmac with a constant: 2 issue clocks; round 4's fma(exit, widen, -entry) held the constant in an SGPR: 4) -- shifted into a mask with one v_alignbit.
#define RL_QSLAB8(wn, wf, sh) { \
	const float nx_ = __builtin_fmaf((float)((nX##wn >> sh) & 0xffu), -Ax_, nBx_), fx_ = __builtin_fmaf((float)((fX##wf >> sh) & 0xffu), Ax_, Bfx_); \
	const float ny_ = __builtin_fmaf((float)((nY##wn >> sh) & 0xffu), -Ay_, nBy_), fy_ = __builtin_fmaf((float)((fY##wf >> sh) & 0xffu), Ay_, Bfy_); \
	const float nz_ = __builtin_fmaf((float)((nZ##wn >> sh) & 0xffu), -Az_, nBz_), fz_ = __builtin_fmaf((float)((fZ##wf >> sh) & 0xffu), Az_, Bfz_); \
	const float ntn_ = fminf(ntMin_, __builtin_fminf(__builtin_fminf(nx_, ny_), nz_)), tf_ = fminf(tmxL_, __builtin_fminf(__builtin_fminf(fx_, fy_), fz_)); \
	culled = __builtin_amdgcn_alignbit(culled, __float_as_uint(__builtin_fmaf(tf_, RL_POOL_WIDEN, ntn_)), 31u); }
#define RL_SEL8(hi_, lo_, m_) __builtin_amdgcn_bitop3_b32((hi_), (lo_), (m_), 0xE4)   /* (hi & m) | (lo & ~m): truth table over (hi, lo, m) */
__device__ __forceinline__ bool NodeStep8(const DSceneView& S, Trav& T, float tMin, int* stk, int* ovf, Counters& c, const unsigned char* perm, const uint4* top, const int G, const int GMAX)
{
	RL_WSTEP(4);
	c.nodes++;   // one 80-byte record
	// the group's next child in visiting order; the rest of the group, if any, is one stack entry
	const uint32_t pos = 31u - (uint32_t)__clz((int)T.gy);
	T.gy &= ~(1u << pos);
	const uint32_t slot = (pos - 24u) ^ T.oct;
	const uint32_t node = T.gx + (uint32_t)__popc(T.gy & 0xffu & ((1u << slot) - 1u));
	if ((T.gy >> 24) != 0u) Push8(T, stk, ovf, G, GMAX);
#ifdef RL_DIAG_TOPN   /* which nodes the steps go to (breadth-first numbers: a prefix is the top of the tree) and how many groups the stack holds: what an LDS copy of the top serves */
	if (c.diag) {
		const uint32_t lim_[8] = { 9u, 22u, 53u, 73u, 128u, 256u, 1024u, 0xffffffffu };
		uint32_t lo_ = 0;
		for (int b_ = 0; b_ < 8; ++b_) { const unsigned long long m_ = Ballot(node >= lo_ && node < lim_[b_]); if (m_ && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)Ballot(1)) - 1u) atomicAdd(&c.diag[CNT_COUNT + 4 + b_], (unsigned long long)__popcll(m_)); lo_ = lim_[b_]; }
		const uint32_t dl_[8] = { 1u, 2u, 3u, 4u, 5u, 6u, 8u, 0xffffffffu };
		lo_ = 0;
		for (int b_ = 0; b_ < 8; ++b_) { const unsigned long long m_ = Ballot((uint32_t)T.sp >= lo_ && (uint32_t)T.sp < dl_[b_]); if (m_ && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)Ballot(1)) - 1u) atomicAdd(&c.diag[CNT_COUNT + 16 + b_], (unsigned long long)__popcll(m_)); lo_ = dl_[b_]; }
	}
#endif
	// five 16-byte rows from global memory: the base is the kernel's (uniform), the offset 32-bit -- global_load with an SGPR base.  (RL_TOP8_NODES > 0, an
	// experiment: the first nodes -- breadth first, the top of the tree -- from the workgroup's LDS copy: rl_device.h.)
	const uint32_t at_ = node * 80u;
	uint4 h_, k_, p0_, p1_, p2_;
#if RL_TOP8_NODES > 0
	if (node < (uint32_t)RL_TOP8_NODES) {
		const char* lp_ = (const char*)top + at_;
		h_ = *(const uint4*)(lp_); k_ = *(const uint4*)(lp_ + 16); p0_ = *(const uint4*)(lp_ + 32); p1_ = *(const uint4*)(lp_ + 48); p2_ = *(const uint4*)(lp_ + 64);
	} else
#endif
	{
		(void)top;
		const char* np_ = (const char*)S.nodes8 + at_;
		h_ = GLoadU4(np_, 0); k_ = GLoadU4(np_, 1); p0_ = GLoadU4(np_, 2); p1_ = GLoadU4(np_, 3); p2_ = GLoadU4(np_, 4);
	}
	const float Ax_ = __uint_as_float((h_.w & 0xffu) << 23) * T.inv.x, Ay_ = __uint_as_float(((h_.w >> 8) & 0xffu) << 23) * T.inv.y, Az_ = __uint_as_float(((h_.w >> 16) & 0xffu) << 23) * T.inv.z;
	const float Bx_ = (__uint_as_float(h_.x) - T.o.x) * T.inv.x, By_ = (__uint_as_float(h_.y) - T.o.y) * T.inv.y, Bz_ = (__uint_as_float(h_.z) - T.o.z) * T.inv.z;
	// (|B| + 255 |A|) * 2^-21, as in RL_WIDE_STEP_Q: four times the rounding of q * A + B against the reference's (bound - o) * inv; -(B - E) and B + E
	const float Ex_ = fabsf(Ax_ * 1.21593475e-4f) + fabsf(Bx_ * 4.76837158e-7f), Ey_ = fabsf(Ay_ * 1.21593475e-4f) + fabsf(By_ * 4.76837158e-7f), Ez_ = fabsf(Az_ * 1.21593475e-4f) + fabsf(Bz_ * 4.76837158e-7f);
	const float nBx_ = Ex_ - Bx_, Bfx_ = Bx_ + Ex_, nBy_ = Ey_ - By_, Bfy_ = By_ + Ey_, nBz_ = Ez_ - Bz_, Bfz_ = Bz_ + Ez_;
	// planes: p0 = qlo x (children 0-3, 4-7), qlo y (0-3, 4-7); p1 = qlo z (0-3, 4-7), qhi x (0-3, 4-7); p2 = qhi y (0-3, 4-7), qhi z (0-3, 4-7)
	const uint32_t nX0 = RL_SEL8(p1_.z, p0_.x, T.m8x), fX0 = RL_SEL8(p0_.x, p1_.z, T.m8x), nX1 = RL_SEL8(p1_.w, p0_.y, T.m8x), fX1 = RL_SEL8(p0_.y, p1_.w, T.m8x);
	const uint32_t nY0 = RL_SEL8(p2_.x, p0_.z, T.m8y), fY0 = RL_SEL8(p0_.z, p2_.x, T.m8y), nY1 = RL_SEL8(p2_.y, p0_.w, T.m8y), fY1 = RL_SEL8(p0_.w, p2_.y, T.m8y);
	const uint32_t nZ0 = RL_SEL8(p2_.z, p1_.x, T.m8z), fZ0 = RL_SEL8(p1_.x, p2_.z, T.m8z), nZ1 = RL_SEL8(p2_.w, p1_.y, T.m8z), fZ1 = RL_SEL8(p1_.y, p2_.w, T.m8z);
	const float ntMin_ = -tMin, tmxL_ = ClampToFltMax(T.best.t);
	uint32_t culled = 0u;   // child 7 first: child c ends up in bit c
	RL_QSLAB8(1, 1, 24) RL_QSLAB8(1, 1, 16) RL_QSLAB8(1, 1, 8) RL_QSLAB8(1, 1, 0)
	RL_QSLAB8(0, 0, 24) RL_QSLAB8(0, 0, 16) RL_QSLAB8(0, 0, 8) RL_QSLAB8(0, 0, 0)
	const uint32_t hitSlot = ~culled & 0xffu;
	// hits -> the next group (inner children, bits moved to visiting order by the workgroup's 8 x 256 table) and the next triangles (leaf children: their bits as
	// they are -- LeafStep8 works out which triangle a bit stands for; round 4 spread every bit into a nibble here, ten instructions on every node step)
	const uint32_t imask = h_.w >> 24;
	const uint32_t innerP = (uint32_t)perm[T.oct * 256u + (hitSlot & imask)];
	T.gx = k_.x; T.gy = (innerP << 24) | imask;
	T.tx = k_.y; T.tz = k_.z; T.ty = (hitSlot & ~imask) | ((k_.w & 0xffu) << 16);
	return Next8(T, stk, ovf, G);
}
template <bool PRIMS>
__device__ __forceinline__ bool LeafStep8(const DSceneView& S, Trav& T, float tMin, int* stk, int* ovf, Counters& c, const int G)
{
	RL_WSTEP(5);
	// the lowest hit leaf child, its next triangle (the children's triangles are consecutive slots: triBase + the bits of leafMask below)
	const uint32_t lc = (uint32_t)__ffs((int)(T.ty & 0xffu)) - 1u;
	const uint32_t k = (T.ty >> 8) & 3u;
	const uint32_t nib = (T.tz >> (4u * lc)) & 15u;
	const int i = (int)(T.tx + (uint32_t)__popc(T.tz & ((1u << (4u * lc)) - 1u)) + k);
	const bool alpha = ((T.ty >> (16u + lc)) & 1u) != 0u;
	if ((nib >> (k + 1u)) != 0u) T.ty += 0x100u;                       // the child has another triangle
	else { T.ty &= ~0x300u; T.ty &= T.ty - 1u; }                       // next child (the lowest set bit is a child's: bits 8 - 9 are clear)
	const V3 o = T.o, d = T.d;
	c.tris++;
	const Tri TT = LoadTri(S, i);
	// reference geom/triangle.cc:22-27
	const float t = dot((TT.v0 - o), TT.n) / dot(d, TT.n);
	if (t >= tMin && t <= FLT_MAX && (t < T.best.t || (t == T.best.t && i < T.best.tri))) {   // ties: the lower slot, as in Traverse()
		const V3 pp = o + t * d;
		const V3 w = pp - TT.v0;
		const float wv = dot(w, TT.v), wu = dot(w, TT.u);
		float pa, pb;
		if (Barycentric(S.fastBary != 0, TT.uv * wv - TT.vv * wu, TT.uv * wu - TT.uu * wv, TT.denom, TT.rden, pa, pb) && OwnBoxPass(TT.v0, TT.v1, TT.v2, o, v3(rtm::rcp1_(d.x), rtm::rcp1_(d.y), rtm::rcp1_(d.z)), tMin, t)) {
			if (!alpha || AlphaTestCandidate(S, i, pa, pb, c)) {
				T.best.t = t; T.best.a = pa; T.best.b = pb; T.best.tri = i;
				if (T.anyhit) return true;
			}
		}
	}
	return Next8(T, stk, ovf, G);
}

// sky part of the miss shader (reference render/renderer.cc:155-181)
__device__ __forceinline__ V3 MissSky(const DSceneView& S, const SkyRot& R, V3 d, Counters& c)
{
	V3 missResult = v3s(0.0f);
	if (S.sky) {
		V3 dir = normalize(d);
		V3 D = v3(dot(ld3(R.m0), dir), dot(ld3(R.m1), dir), dot(ld3(R.m2), dir));
		float u = rtm::atan2_(D.z, D.x), v = rtm::asin_(D.y);
		u *= 0.1591f; v *= 0.3183f;
		u += 0.5f; v += 0.5f;
		int x = (int)(u * (float)(uint32_t)(S.skyWidth - 1));
		int y = (int)(v * (float)(uint32_t)(S.skyHeight - 1));
		float4 px = ((const float4*)S.sky)[(uint32_t)(y * S.skyWidth + x)];
		c.texels++;
		missResult = missResult + v3(px.x, px.y, px.z);
	}
	return missResult;
}

// Radiance folded from the last vertex back to the camera: radiance = (0 + refl*Li*sp/pdf) + E at every vertex,
// in the reference's operation order (renderer.cc:139-151).
__device__ __forceinline__ V3 FoldPath(const float* __restrict__ pathStack, uint32_t stackStride, uint32_t home, int depth, V3 L)
{
#if RL_FOLD_PREFETCH_POOL > 0
	if (depth <= RL_FOLD_PREFETCH_POOL) {
		float4 q0[RL_FOLD_PREFETCH_POOL], q1[RL_FOLD_PREFETCH_POOL];
		#pragma unroll
		for (int k = 0; k < RL_FOLD_PREFETCH_POOL; ++k) {
			const int kk = k < depth ? k : 0;
			const float4* rec = (const float4*)pathStack + ((size_t)kk * stackStride + home) * 2u;
			q0[k] = rec[0]; q1[k] = rec[1];
		}
		#pragma unroll
		for (int k = RL_FOLD_PREFETCH_POOL - 1; k >= 0; --k) {
			if (k < depth) {
				const V3 refl = v3(q0[k].x, q0[k].y, q0[k].z);
				const float sp = q0[k].w, pdf = q1[k].x;
				const V3 E = v3(q1[k].y, q1[k].z, q1[k].w);
				V3 radiance = v3s(0.0f);
				radiance = radiance + refl * L * sp / pdf;
				radiance = radiance + E;
				L = radiance;
			}
		}
		return L;
	}
#endif
	for (int k = depth - 1; k >= 0; --k) {
		const float4* rec = (const float4*)pathStack + ((size_t)k * stackStride + home) * 2u;
		const float4 r0 = rec[0], r1 = rec[1];
		const V3 refl = v3(r0.x, r0.y, r0.z);
		const float sp = r0.w, pdf = r1.x;
		const V3 E = v3(r1.y, r1.z, r1.w);
		V3 radiance = v3s(0.0f);
		radiance = radiance + refl * L * sp / pdf;
		radiance = radiance + E;
		L = radiance;
	}
	return L;
}

template <int LSTACK, bool PRIMS, int K> struct PoolOcc {
	static constexpr int kFields = PRIMS ? F_COUNT : F_COUNT - 1;
	static constexpr int kLdsPerBlock = LSTACK * RL_BLOCK * 4 + (RL_BLOCK / 64) * (kFields * 64 * K * 4 + 64 * K);
	static constexpr int kFit = (160 * 1024) / kLdsPerBlock;
	static constexpr int kBlocks = kFit < 1 ? 1 : (kFit > RL_POOL_MAXBLOCKS ? RL_POOL_MAXBLOCKS : kFit);
};

// STACK: capacity of the traversal stack; LSTACK <= STACK: how much of it lives in LDS (the rest is private overflow)
// WIDE: 0 the BVH2; 1 the BVH4 (S.nodes4: 64-byte grid nodes; RL_Q4 = 0: float boxes); 3 the 8-wide tree (S.nodes8; STACK / LSTACK then count words: two per group)
template <int STACK, bool PRIMS, int K, int LSTACK = STACK, int WIDE = 0>
__global__ void __launch_bounds__(RL_BLOCK, (PoolOcc<LSTACK, PRIMS, K>::kBlocks))
k_trace_pool(const DRenderParams Pk, const DSceneView Sk, const SkyRot Rk, SampleRGB* __restrict__ samplesK,
             float* __restrict__ pathStackK, unsigned long long* __restrict__ countersK, unsigned int* __restrict__ jobCounterK)
#ifndef RL_TU_POOL
;   // defined in the translation unit of rl_render_pool.hip: this same source, compiled with a scheduler strategy of its own (Makefile); instances below
#else
{
	(void)Pk; (void)Sk; (void)Rk; (void)samplesK; (void)pathStackK; (void)countersK; (void)jobCounterK;   // read through RL_ARGS() where a part of the loop needs them (k_trace)
	RL_TEX_PROLOGUE(Sk);
	RL_MATH_PROLOGUE();
	constexpr int PP = 64 * K;
	static_assert(LSTACK <= STACK, "the LDS part cannot exceed the stack");
	__shared__ int s_stack[LSTACK * RL_BLOCK];
	int ovfStore[LSTACK < STACK ? STACK - LSTACK : 1];
	int* ovf = ovfStore;
	__shared__ float s_pool[RL_BLOCK / 64][PoolOcc<LSTACK, PRIMS, K>::kFields][PP];
	__shared__ unsigned char s_free[RL_BLOCK / 64][PP];
	// the 8-wide walk: s_perm[oct * 256 + y] = the byte y with every bit b moved to bit b XOR oct (slot order -> visiting order of a ray of octant oct)
	// ... and the top of that tree: its first RL_TOP8_NODES nodes (rl_device.h), five 16-byte rows each
	__shared__ uint4 s_top[(WIDE == 3 && RL_TOP8_NODES > 0) ? RL_TOP8_NODES * 5 : 1];
	__shared__ unsigned char s_perm[WIDE == 3 ? 8 * 256 : 1];
	if constexpr (WIDE == 3) {
#if RL_TOP8_NODES > 0
		{
			RL_ARGS();
			const uint32_t rows = (uint32_t)(S.numNodes8 < RL_TOP8_NODES ? S.numNodes8 : RL_TOP8_NODES) * 5u;
			for (uint32_t i = threadIdx.x; i < (uint32_t)RL_TOP8_NODES * 5u; i += RL_BLOCK) s_top[i] = i < rows ? GLoadU4(S.nodes8, (int)i) : make_uint4(0u, 0u, 0u, 0u);
		}
#endif
		for (uint32_t i = threadIdx.x; i < 8u * 256u; i += RL_BLOCK) {
			const uint32_t m = i >> 8, y = i & 255u;
			uint32_t r = 0;
			for (uint32_t bb = 0; bb < 8u; ++bb) if ((y >> bb) & 1u) r |= 1u << (bb ^ m);
			s_perm[i] = (unsigned char)r;
		}
		__syncthreads();
	}
	constexpr int G8 = LSTACK / 2, GMAX8 = STACK / 2;   // groups in the LDS part of the stack, groups in all

	int* stk = s_stack + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	float (*pool)[PP] = s_pool[wave];
	unsigned char* freeList = s_free[wave];
	uint32_t numSlots;
	JobSource js;
	{ RL_ARGS(); numSlots = P.numLocalCells * 64u; js = JobSourceInit(P); }
	const unsigned long long laneLt = (1ull << lane) - 1ull;
	// path-stack column of home slot p: consecutive lanes -> consecutive columns
	const uint32_t homeBase = blockIdx.x * (RL_BLOCK * K) + threadIdx.x;

	Counters c; c.rays = c.nodes = c.tris = c.shaded = c.texels = c.samples = c.trips = 0; RL_DIAG_BIND(c);
	// home-lane registers of slot p*64 + lane
	unsigned long long stRng[K];
	uint32_t stOut[K];
	int stDepth[K];
	bool stActive[K];
	#pragma unroll
	for (int p = 0; p < K; ++p) { stRng[p] = 0; stOut[p] = 0; stDepth[p] = 0; stActive[p] = false; pool[F_TRI][p * 64 + lane] = __int_as_float(Q_EMPTY); }
	// (Round 2 gave every wave its first chunk without an atomic, because 4096 waves asking ONE counter at the same instant stood in line for ~45 us; with a
	// head per XCD the line is an eighth as long and the first chunk comes from the wave's own band like every other.)
	uint32_t chunkNext = 0, chunkEnd = 0;
	bool globalDone = false, exhausted = false;   // wave-uniform
#ifdef RL_DIAG_TIMELINE
	const uint32_t gtid = blockIdx.x * RL_BLOCK + threadIdx.x;
#endif
	RL_TIMELINE(0);
	uint32_t surviveQ8 = 256u;                    // share of freshly generated camera samples that reached a pool slot, x 256 (wave-uniform)
	// traversal state of the ray this lane is tracing; survives trips (a straggler keeps going while the rest of the pool is shaded)
	// A lane without a ray has T.cur == IDLE (no node index, not negative like a leaf reference): "busy", "at an inner node", "at a leaf" are then ONE integer
	// compare each, and a wave vote on a compare is that compare's lane mask.  (A vote on a bool that is not a compare -- `busy && T.cur >= 0` -- makes the
	// compiler write the bool out as 0 / 1 and compare it with zero again: v_cndmask + v_cmp_ne per vote, four votes per traversal step.)
	constexpr int IDLE = 0x7fffffff;
	int mySlot = 0;
	Trav T;
	T.o = T.d = T.inv = v3s(0.0f); T.rayTime = 0.0f; T.nx = T.ny = T.nz = T.anyhit = false;
	T.best.t = INFINITY; T.best.a = T.best.b = 0.0f; T.best.tri = -1; T.cur = IDLE; T.sp = 0; T.leafI = 0;
	T.gx = T.gy = T.tx = T.ty = T.tz = T.oct = 0u; T.m8x = T.m8y = T.m8z = 0u;
#ifdef RL_DIAG_STAMPS
	unsigned long long stampAcc[4] = { 0, 0, 0, 0 };
	{ RL_ARGS(); c.diag = counters; }
	unsigned long long stampLast = __builtin_amdgcn_s_memtime();
	#define RL_PSTAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stampAcc[k] += now_ - stampLast; stampLast = now_; __builtin_amdgcn_sched_barrier(0); }
#else
	#define RL_PSTAMP(k)
#endif

#ifdef RL_POOL_WATCHDOG
	uint32_t wdSteps = 0, wdTrips = 0; bool wdAbort = false;
#endif
	for (;;) {
#ifdef RL_POOL_WATCHDOG
		if (++wdTrips > 20000u || Ballot(wdAbort) != 0ull) {
			uint32_t nAct = 0, nEmpty = 0, nQ = 0, nH = 0, nPend = 0, nRes = 0;
			for (int p = 0; p < K; ++p) {
				const int q = __float_as_int(pool[F_TRI][p * 64 + (int)lane]);
				nAct += (uint32_t)__popcll(Ballot(stActive[p]));
				nEmpty += (uint32_t)__popcll(Ballot(stActive[p] && q == Q_EMPTY));
				nQ += (uint32_t)__popcll(Ballot(stActive[p] && (q == Q_CLOSEST || q == Q_SHADOW)));
				nH += (uint32_t)__popcll(Ballot(stActive[p] && q >= 0));
				nPend += (uint32_t)__popcll(Ballot(stActive[p] && (q == Q_PENDING || q == Q_PENDING_SHADOW)));
				nRes += (uint32_t)__popcll(Ballot(stActive[p] && (q == Q_MISS || q == Q_CLEAR || q == Q_OCCLUDED)));
			}
			if (lane == 0) {
				RL_ARGS();
				atomicAdd(&counters[CNT_COUNT + 21], 1ull);
				atomicAdd(&counters[CNT_COUNT + 4], (unsigned long long)nAct); atomicAdd(&counters[CNT_COUNT + 5], (unsigned long long)nEmpty);
				atomicAdd(&counters[CNT_COUNT + 6], (unsigned long long)nQ); atomicAdd(&counters[CNT_COUNT + 7], (unsigned long long)nH);
				atomicAdd(&counters[CNT_COUNT + 8], (unsigned long long)nPend); atomicAdd(&counters[CNT_COUNT + 9], (unsigned long long)nRes);
				atomicAdd(&counters[CNT_COUNT + 10], (unsigned long long)(exhausted ? 1 : 0)); atomicAdd(&counters[CNT_COUNT + 11], (unsigned long long)__popcll(Ballot(T.cur != IDLE)));
			}
			break;
		}
#endif
		// ---- refill: deal new camera samples to the free slots (wave64 ballot + prefix ranks) ----
		if (!exhausted) {
			RL_ARGS();
			uint32_t pos[K];
			uint32_t nFree = 0;
			#pragma unroll
			for (int p = 0; p < K; ++p) {
				const bool fr = !stActive[p];
				const unsigned long long m = Ballot(fr);
				pos[p] = fr ? nFree + (uint32_t)__popcll(m & laneLt) : 0xffffffffu;
				if (fr) freeList[pos[p]] = (unsigned char)(p * 64 + (int)lane);
				nFree += (uint32_t)__popcll(m);
			}
			WaveLdsSync();
			uint32_t filled = 0;
			for (int round = 0; round < RL_REFILL_ROUNDS && filled < nFree; ++round) {
				RL_WSTEP(6);   // (level-2 diagnostic build: refill rounds, wave level -- tools/dynamic_mix.py)
				if (chunkNext >= chunkEnd && !globalDone) {
					uint32_t base = 0, bend = 0;
					// (a chunk shared by the workgroup's waves in 64-job batches, as in the leaf-list kernel, was measured here too: 44.5 ms against 43.9)
					if (!TakeJobs(P, jobCounter, js, P.jobChunk, lane, base, bend)) { globalDone = true; RL_TIMELINE(1); }
					else { chunkNext = base; chunkEnd = bend; }
				}
				const uint32_t avail = chunkEnd - chunkNext;
				if (avail == 0) { exhausted = true; break; }
				// How many camera samples to generate this round.  A sample that misses the scene's root box is finished right here and
				// fills no slot; where most do (a camera outside the model: 90 % in the configs[2] stand-in) asking for exactly as many
				// samples as there are free slots fills a tenth of them per round.  So the round asks for more -- free slots / the share
				// that survived lately -- and, if more survive than fit, keeps the first `room` survivors and hands the jobs behind the
				// last one kept back to the queue (chunkNext only advances past the lanes that were committed: the same jobs come
				// round again, same pixel, same stream).  Nothing is written or counted for a lane before it is committed.
				const uint32_t room = nFree - filled;
				uint32_t want = room;
				if (surviveQ8 < 230u) want = min(64u, (room * 256u) / max(24u, surviveQ8 + (surviveQ8 >> 3)));   // a little under 1 / survival rate
				const uint32_t take = min(min(64u, max(room, want)), avail);
				bool alive = false, quick = false;   // quick: decided by the root test (sample written at commit)
				V3 o = v3s(0.0f), d = v3s(0.0f);
				float rayTime = 0.0f;
				Rng g; g.s.state = 0;
				uint32_t outIndex = 0;
				if (lane < take) {
					const JobPixel j = DecodeJob(P, chunkNext + lane);
					if (j.valid) {
						// GenerateCell body, reference render/renderer.cc:232-239
						const uint32_t sidx = P.sampleBegin + j.sample;
						g.s = raylib_rng_begin_mixed(P.seedMixed, j.y * P.width + j.x, sidx);
						float u, v;
						PixelUV(P, j.x, j.y, sidx, g, u, v);
						CameraRay(P.camera, u, v, g, o, d, rayTime);
						outIndex = j.sample * numSlots + j.slot;
						alive = true;
						if (P.maxPathLength <= 0) { alive = false; quick = true; }   // renderer.cc:120-123
						else if (RootMiss(S, o, d, P.rayTMin)) {
							// cannot hit anything: sky lookup plus (with a sun) one occlusion query that may be decided at the root too
							const bool sunQuick = !S.hasSun || RootMiss(S, o, -ld3(S.sunDirection), P.rayTMin);
							if (sunQuick) { alive = false; quick = true; }
						}
					}
				}
				unsigned long long am = Ballot(alive);
				uint32_t n = (uint32_t)__popcll(am);
				uint32_t commit = take;                    // lanes [0, commit) are this round's samples
				if (n > room) {
					// the lane of the (room + 1)-th survivor: everything from there on goes back to the queue
					const unsigned long long over = Ballot(alive && (uint32_t)__popcll(am & laneLt) == room);
					commit = (uint32_t)__ffsll((long long)over) - 1u;
					if (lane >= commit) { alive = false; quick = false; }
					am = Ballot(alive);
					n = room;
				}
				{   // survival rate of the committed samples, 8-bit fixed point, smoothed over the last few rounds
					const uint32_t rate = commit ? (n * 256u) / commit : 256u;
					surviveQ8 = (surviveQ8 * 3u + rate + 2u) >> 2;
				}
				if (lane < commit && (alive || quick)) c.samples++;
				if (quick) {
					if (P.maxPathLength <= 0) samples[outIndex] = make_sample(0.0f, 0.0f, 0.0f);
					else {
						c.rays++; c.nodes++;
						V3 L = MissSky(S, R, d, c);
						if (S.hasSun) { c.rays++; c.nodes++; L = L + ld3(S.sunIlluminance); }
						samples[outIndex] = make_sample(L.x, L.y, L.z);
					}
				}
				chunkNext += commit;
				if (n == 0) continue;
				if (alive) {
					// the r-th surviving ray goes to the (filled + r)-th free slot; the fields a traversal fills in later carry
					// the RNG state and the output index to the slot's home lane
					const int f = (int)freeList[filled + (uint32_t)__popcll(am & laneLt)];
					pool[F_OX][f] = o.x; pool[F_OY][f] = o.y; pool[F_OZ][f] = o.z;
					pool[F_DX][f] = d.x; pool[F_DY][f] = d.y; pool[F_DZ][f] = d.z;
					if (PRIMS) pool[F_TIME][f] = rayTime;
					pool[F_TRI][f] = __int_as_float(Q_CLOSEST);
					pool[F_T][f] = __int_as_float((int)(uint32_t)(g.s.state & 0xffffffffull));
					pool[F_A][f] = __int_as_float((int)(uint32_t)(g.s.state >> 32));
					pool[F_B][f] = __int_as_float((int)outIndex);
				}
				WaveLdsSync();
				#pragma unroll
				for (int p = 0; p < K; ++p) {
					if (pos[p] >= filled && pos[p] < filled + n) {
						const int slot = p * 64 + (int)lane;
						stRng[p] = (unsigned long long)(uint32_t)__float_as_int(pool[F_T][slot]) | ((unsigned long long)(uint32_t)__float_as_int(pool[F_A][slot]) << 32);
						stOut[p] = (uint32_t)__float_as_int(pool[F_B][slot]);
						stDepth[p] = 0;
						stActive[p] = true;
					}
				}
				filled += n;
			}
		}
		bool anyActive = false;
		#pragma unroll
		for (int p = 0; p < K; ++p) anyActive = anyActive || stActive[p];
		if (Ballot(anyActive) == 0ull) {
			if (exhausted) break;
			continue;
		}
		if (lane == 0) c.trips++;
		RL_PSTAMP(0);

		// ---- traversal phase: every waiting query of the pool; a lane takes the next slot whenever its ray is finished ----
		{
			RL_ARGS();
			const float tMinC = __builtin_canonicalizef(P.rayTMin);   // known to be canonical: the box tests' max chains start from it without a v_max x, x per step
#if RL_POOL_NODEPTR_VGPR
			// the wide nodes' base address in a VGPR pair for the phase: as one of the loop's many uniform values it would be spilled to a VGPR's lanes and
			// read back (two v_readlane, 4 issue cycles each) at every traversal step
			DSceneView St = S;
			if constexpr (WIDE == 3) { }   // (the 8-wide node's rows are loaded from an SGPR base plus a 32-bit offset: NodeStep8)
			else { const DNode4Q* pn = S.nodes4; asm volatile("" : "+v"(pn)); St.nodes4 = pn; }
#else
			const DSceneView& St = S;
#endif
			WaveLdsSync();
			uint32_t nextSlot = 0;
			uint32_t finished = 0;      // rays completed in this phase (wave-uniform)
			for (;;) {
				if (nextSlot < (uint32_t)PP) {
					const unsigned long long idle = Ballot(T.cur == IDLE);
					const uint32_t slot = nextSlot + (uint32_t)__popcll(idle & laneLt);
					if (T.cur == IDLE && slot < (uint32_t)PP) {
						const int q = __float_as_int(pool[F_TRI][slot]);
						if (q == Q_CLOSEST || q == Q_SHADOW) {
							T.o = v3(pool[F_OX][slot], pool[F_OY][slot], pool[F_OZ][slot]);
							T.anyhit = (q == Q_SHADOW);
							T.d = T.anyhit ? -ld3(S.sunDirection) : v3(pool[F_DX][slot], pool[F_DY][slot], pool[F_DZ][slot]);
							T.rayTime = PRIMS ? pool[F_TIME][slot] : 0.0f;
							T.inv = v3(FastRcp(T.d.x), FastRcp(T.d.y), FastRcp(T.d.z));
							if ((WIDE && RL_Q4) || WIDE == 3) T.inv = ClampInv(T.inv);   // only the grid nodes' fused plane arithmetic wants finite reciprocals; Slab() relies on +-inf / NaN
							T.nx = T.inv.x < 0.0f; T.ny = T.inv.y < 0.0f; T.nz = T.inv.z < 0.0f;
							T.best.t = INFINITY; T.best.tri = -1; T.best.a = 0.0f; T.best.b = 0.0f;
							T.cur = 0; T.sp = 0; T.leafI = 0;
							if constexpr (WIDE == 3) {
								// the root as a group of one: base 0, imask 1, its bit at the visiting position of slot 0
								RaySetup8(T);
								T.gx = 0u; T.gy = (1u << (24u + T.oct)) | 1u; T.tx = T.ty = T.tz = 0u;
							}
							mySlot = (int)slot;
							pool[F_TRI][slot] = __int_as_float(T.anyhit ? Q_PENDING_SHADOW : Q_PENDING);
							c.rays++;
						}
					}
					nextSlot += (uint32_t)__popcll(idle);
				}
				const int nBusy = (int)__popcll(Ballot(T.cur != IDLE));
				if (nBusy == 0) {
					if (nextSlot >= (uint32_t)PP) break;
					continue;
				}
				// all queries handed out and only a few long rays left: shade what is there, the stragglers go on next trip
				const int cutAt = exhausted ? RL_POOL_CUT_EXH : RL_POOL_CUT;
				if (nextSlot >= (uint32_t)PP && nBusy <= cutAt && finished > 0) break;
				// one step for the larger (cost-weighted) party, lanes at inner nodes or lanes at leaves, until enough lanes
				// have finished to make a fetch worth it
				int nb;
				do {
					const bool atNode = (uint32_t)T.cur < (uint32_t)IDLE, atLeaf = T.cur < 0;
					const int nN = (int)__popcll(Ballot(atNode)), nL = (int)__popcll(Ballot(atLeaf));
					bool fin = false;
					const bool nodeTurn = nN * (WIDE == 3 ? RL_POOL_WNODE8 : WIDE ? RL_POOL_WNODE4 : RL_POOL_WNODE) >= nL * (WIDE == 3 ? RL_POOL_WLEAF8 : WIDE ? RL_POOL_WLEAF4 : RL_POOL_WLEAF);
					if constexpr (WIDE == 3) {
#if RL_POOL_BOTH8
						// both parties every turn: a lane at a node takes its node step, a lane at a leaf its triangle step (the wave runs either part only if some lane needs it)
						(void)nodeTurn;
						if (atNode) fin = NodeStep8(St, T, tMinC, stk, ovf, c, s_perm, s_top, G8, GMAX8);
						else if (atLeaf) fin = LeafStep8<PRIMS>(S, T, P.rayTMin, stk, ovf, c, G8);
#else
						if (nodeTurn) { if (atNode) fin = NodeStep8(St, T, tMinC, stk, ovf, c, s_perm, s_top, G8, GMAX8); }
						else { if (atLeaf) fin = LeafStep8<PRIMS>(S, T, P.rayTMin, stk, ovf, c, G8); }
#endif
					} else {
						if (nodeTurn) { if (atNode) fin = WIDE ? NodeStep4<LSTACK, STACK>(St, T, tMinC, stk, ovf, c) : NodeStep<LSTACK, STACK>(S, T, tMinC, stk, ovf, c); }
						else { if (atLeaf) fin = LeafStep<LSTACK, STACK, PRIMS>(S, T, P.rayTMin, stk, ovf, c); }
					}
					if (fin) {
						const bool hit = T.best.tri >= 0;
						int q = T.best.tri;
						if (T.anyhit) q = hit ? Q_OCCLUDED : Q_CLEAR;
						else if (!hit) q = Q_MISS;
						pool[F_T][mySlot] = T.best.t; pool[F_TRI][mySlot] = __int_as_float(q);
						pool[F_A][mySlot] = T.best.a; pool[F_B][mySlot] = T.best.b;
						T.cur = IDLE;
					}
#ifdef RL_POOL_WATCHDOG
					if (++wdSteps > 400000u) { if (lane == 0) atomicAdd(&counters[CNT_COUNT + 20], 1ull); T.cur = IDLE; wdAbort = true; }
#endif
					nb = (int)__popcll(Ballot(T.cur != IDLE));
					finished += (uint32_t)(nN + nL - nb);   // whoever was busy and is not any more has finished its ray
				} while (nb > (nextSlot < (uint32_t)PP ? RL_POOL_KEEP : (finished > 0 ? cutAt : 0)));
			}
			WaveLdsSync();
		}
		RL_PSTAMP(1);

		// ---- shading (TraceScene after the accel->Hit call, reference render/renderer.cc:129-208) ----
		// (1) the cheap outcomes are finished by the slot's home lane: a miss runs the sky lookup and (with a sun) turns
		//     into an occlusion query, a returned occlusion query ends the path.  Hits are only LISTED.
		uint32_t nHit = 0;
		uint32_t hitIdx[K];
		{
		RL_ARGS();
		#pragma unroll
		for (int p = 0; p < K; ++p) {
			const int slot = p * 64 + (int)lane;
			const int q = __float_as_int(pool[F_TRI][slot]);
			const bool isHit = stActive[p] && q >= 0;
			if (stActive[p] && (q == Q_MISS || q == Q_CLEAR || q == Q_OCCLUDED)) {
				const V3 d = v3(pool[F_DX][slot], pool[F_DY][slot], pool[F_DZ][slot]);
				bool done = true;
				V3 L;
				if (q == Q_MISS) {
					L = MissSky(S, R, d, c);
					if (S.hasSun) {
						// the sky part waits in the direction fields (the sun query brings its own direction)
						pool[F_DX][slot] = L.x; pool[F_DY][slot] = L.y; pool[F_DZ][slot] = L.z;
						pool[F_TRI][slot] = __int_as_float(Q_SHADOW);
						done = false;
					}
				} else {
					L = d;
					if (q == Q_CLEAR) L = L + ld3(S.sunIlluminance);
				}
				if (done) {
					L = FoldPath(pathStack, P.stackStride, homeBase + (uint32_t)p * RL_BLOCK, stDepth[p], L);
					samples[stOut[p]] = make_sample(L.x, L.y, L.z);
					stActive[p] = false;
					pool[F_TRI][slot] = __int_as_float(Q_EMPTY);
				}
			}
			const unsigned long long hm = Ballot(isHit);
			hitIdx[p] = isHit ? nHit + (uint32_t)__popcll(hm & laneLt) : 0xffffffffu;
			if (isHit) freeList[hitIdx[p]] = (unsigned char)slot;
			nHit += (uint32_t)__popcll(hm);
		}
		}
		WaveLdsSync();
		// (2) hits are shaded 64 at a time by whichever lane: the expensive material code always runs with a full wave.
		//     A remainder below 64 waits in its slots for the next trip's hits (until the job queue is empty).
		//     The path registers come from the home lane by ds_bpermute and return through the slot's hit fields.
		uint32_t shadedEnd = 0;
		for (;;) {
			RL_ARGS();
			if (shadedEnd >= nHit) break;
			if (nHit - shadedEnd < (uint32_t)RL_POOL_SHADE_MIN && !exhausted) break;   // once the job queue is empty no refill will top the list up: waiting only stretches the tail
#ifdef RL_POOL_WATCHDOG
			if (++wdSteps > 400000u) { if (lane == 0) atomicAdd(&counters[CNT_COUNT + 20], 1ull); wdAbort = true; break; }
#endif
			RL_WSTEP(7);   // (level-2 diagnostic build: rounds of hit shading, wave level)
			const uint32_t idx = shadedEnd + lane;
			const bool on = idx < nHit;
			const int slot = on ? (int)freeList[idx] : 0;
			const int h = slot & 63, pp = slot >> 6;
			uint32_t rngLo = 0, rngHi = 0, outIndex = 0; int depth = 0;
			#pragma unroll
			for (int k = 0; k < K; ++k) {
				const uint32_t a0 = (uint32_t)__shfl((int)(uint32_t)(stRng[k] & 0xffffffffull), h);
				const uint32_t a1 = (uint32_t)__shfl((int)(uint32_t)(stRng[k] >> 32), h);
				const uint32_t a2 = (uint32_t)__shfl((int)stOut[k], h);
				const int a3 = __shfl(stDepth[k], h);
				if (pp == k) { rngLo = a0; rngHi = a1; outIndex = a2; depth = a3; }
			}
			if (on) {
				const V3 o = v3(pool[F_OX][slot], pool[F_OY][slot], pool[F_OZ][slot]);
				const V3 d = v3(pool[F_DX][slot], pool[F_DY][slot], pool[F_DZ][slot]);
				HitRec hr; hr.t = pool[F_T][slot]; hr.tri = __float_as_int(pool[F_TRI][slot]); hr.a = pool[F_A][slot]; hr.b = pool[F_B][slot];
				const uint32_t home = blockIdx.x * (RL_BLOCK * K) + (uint32_t)pp * RL_BLOCK + wave * 64u + (uint32_t)h;
				Rng g; g.s.state = (unsigned long long)rngLo | ((unsigned long long)rngHi << 32);
				Surf sf;
				const Mat m = LoadMat(S, BuildSurface<PRIMS>(S, o, d, hr, sf, true, c));
				V3 refl = v3s(0.0f), outD = v3s(0.0f);
				float pdf = 0.0f, sp = 0.0f;
				const bool scattered = Scatter(S, m, d, sf, g, c, refl, outD, pdf, sp);
				const V3 E = Emitted(S, m, sf, c);
				bool done = false;
				V3 L = v3s(0.0f);
				if (scattered && pdf > 0.0f) {
					float4* rec = (float4*)pathStack + ((size_t)depth * P.stackStride + home) * 2u;
					rec[0] = make_float4(refl.x, refl.y, refl.z, sp);
					rec[1] = make_float4(pdf, E.x, E.y, E.z);
					depth++;
					if (depth >= P.maxPathLength) done = true;   // the next TraceScene returns 0 at once (renderer.cc:120-123)
					else {
						pool[F_OX][slot] = sf.p.x; pool[F_OY][slot] = sf.p.y; pool[F_OZ][slot] = sf.p.z;
						pool[F_DX][slot] = outD.x; pool[F_DY][slot] = outD.y; pool[F_DZ][slot] = outD.z;
						pool[F_TRI][slot] = __int_as_float(Q_CLOSEST);
					}
				} else {
					L = v3s(0.0f) + E;                            // radiance(0) += Emitted, renderer.cc:137,151
					done = true;
				}
				if (done) {
					L = FoldPath(pathStack, P.stackStride, home, depth, L);
					samples[outIndex] = make_sample(L.x, L.y, L.z);
					pool[F_TRI][slot] = __int_as_float(Q_EMPTY);
				}
				// back to the home lane: RNG state and depth (negative = the path has ended)
				pool[F_T][slot] = __int_as_float((int)(uint32_t)(g.s.state & 0xffffffffull));
				pool[F_A][slot] = __int_as_float((int)(uint32_t)(g.s.state >> 32));
				pool[F_B][slot] = __int_as_float(done ? -1 : depth);
			}
			shadedEnd += 64u;
		}
		WaveLdsSync();
		// (3) the home lanes take their registers back
		#pragma unroll
		for (int p = 0; p < K; ++p) {
			if (hitIdx[p] < shadedEnd) {
				const int slot = p * 64 + (int)lane;
				stRng[p] = (unsigned long long)(uint32_t)__float_as_int(pool[F_T][slot]) | ((unsigned long long)(uint32_t)__float_as_int(pool[F_A][slot]) << 32);
				const int dd = __float_as_int(pool[F_B][slot]);
				if (dd < 0) stActive[p] = false; else stDepth[p] = dd;
			}
		}
		RL_PSTAMP(2);
	}

	RL_ARGS();
#ifdef RL_DIAG_STAMPS
	if (lane == 0) for (int k = 0; k < 4; ++k) { atomicAdd(&counters[CNT_COUNT + k], stampAcc[k]); atomicAdd(&counters[CNT_COUNT + 12 + k], c.tAcc[k]); }
#endif
	RL_TIMELINE(2);
	uint32_t vals[CNT_COUNT] = { c.rays, c.nodes, c.tris, c.shaded, c.texels, c.samples, c.trips };
	for (int k = 0; k < CNT_COUNT; ++k) {
		unsigned long long v = vals[k];
		for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
		if (lane == 0 && v) atomicAdd(&counters[k], v);
	}
}
#endif
// The instances the runtime selects from (rl_runtime.inl SelectTraceKernel): defined in rl_render_pool.hip's translation unit, referenced from this one.
#define RL_POOL_INSTANCES(X) \
	X(16, false, 2, 16, 0) X(16, false, 3, 16, 0) X(16, false, 4, 16, 0) X(32, false, 2, 32, 0) X(32, false, 3, 32, 0) X(32, false, 4, 32, 0) \
	X(32, false, 2, 4, 0) X(32, false, 2, RL_POOL_SHORT_LSTACK, 0) \
	X(32, false, 2, 32, 1) X(64, false, 2, 32, 1) X(32, false, 2, RL_POOL_SHORT_LSTACK, 1) X(64, false, 2, RL_POOL_SHORT_LSTACK, 1) \
	X(2 * RL_POOL8_MAXLEVELS, false, 2, RL_POOL8_LSTACK, 3)
#ifdef RL_TU_POOL
#define RL_POOL_X(a, b, c, d, e) template __global__ void k_trace_pool<a, b, c, d, e>(const DRenderParams, const DSceneView, const SkyRot, SampleRGB* __restrict__, float* __restrict__, unsigned long long* __restrict__, unsigned int* __restrict__);
#else
#define RL_POOL_X(a, b, c, d, e) extern template __global__ void k_trace_pool<a, b, c, d, e>(const DRenderParams, const DSceneView, const SkyRot, SampleRGB* __restrict__, float* __restrict__, unsigned long long* __restrict__, unsigned int* __restrict__);
#endif
RL_POOL_INSTANCES(RL_POOL_X)
#undef RL_POOL_X

#ifndef RL_TU_POOL   // everything below belongs to the main translation unit alone
// Sequential per-pixel sum of this batch's samples, then (last batch) the mean.
// reference render/renderer.cc:244-248 + core/vec3.h:214-220 (operator/= multiplies by 1/SPP)
__global__ void __launch_bounds__(RL_BLOCK)
k_resolve(const DRenderParams P, const DSceneView S, const SkyRot R, const SampleRGB* __restrict__ samples, float4* __restrict__ accum, float4* __restrict__ out, int firstBatch, int lastBatch)
{
	RL_MATH_PROLOGUE();
	const uint32_t numSlots = P.numLocalCells * 64u;
	const uint32_t slot = blockIdx.x * RL_BLOCK + threadIdx.x;
	if (slot >= numSlots) return;
	const uint32_t p = slot & 63u, cellLocal = slot >> 6;
	const uint32_t cell = P.cellFirst + cellLocal * P.cellStride;
	const uint32_t x = (cell % P.cellsX) * 8u + (p & 7u), y = (cell / P.cellsX) * 8u + (p >> 3);
	const bool valid = x < P.width && y < P.height;
	float4 a = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
	if (valid) {
		if (!firstBatch) a = accum[slot];
		if (P.cellEmpty && P.cellEmpty[cellLocal]) {
			// a cell outside the scene's silhouette (rl_cull.cc): none of its samples can meet the scene, every one of them is the miss shader's value -- the sun's
			// illuminance or nothing, the same for all; with a sky panorama the texel its camera ray points at on top (renderer.cc:155-199), so the ray is
			// generated here exactly as the megakernel generates it (same stream, same draws: jitter, lens, shutter) -- added up sample by sample as if stored
			if (P.emptySky) {
				Counters c; c.rays = c.nodes = c.tris = c.shaded = c.texels = c.samples = c.trips = 0; RL_DIAG_BIND(c);
				for (uint32_t s = 0; s < P.sampleCount; ++s) {
					const uint32_t sidx = P.sampleBegin + s;
					Rng g; g.s = raylib_rng_begin_mixed(P.seedMixed, y * P.width + x, sidx);
					float u, v;
					PixelUV(P, x, y, sidx, g, u, v);
					V3 o, d; float rayTime;
					CameraRay(P.camera, u, v, g, o, d, rayTime);
					V3 L = MissSky(S, R, d, c);
					if (S.hasSun) L = L + ld3(S.sunIlluminance);
					a.x += L.x; a.y += L.y; a.z += L.z;
				}
			} else
			for (uint32_t s = 0; s < P.sampleCount; ++s) { a.x += P.emptyL[0]; a.y += P.emptyL[1]; a.z += P.emptyL[2]; }
		} else
		for (uint32_t s = 0; s < P.sampleCount; ++s) {
			const SampleRGB v = samples[(size_t)s * numSlots + slot];
			a.x += v.x; a.y += v.y; a.z += v.z;
		}
		if (lastBatch) {
			const float k = rtm::rcp1_((float)P.spp);
			a.x *= k; a.y *= k; a.z *= k; a.w = 1.0f;
		} else {
			accum[slot] = a;
		}
	}
	if (lastBatch) {
		if (P.rowMajorOutput) { if (valid) out[(size_t)y * P.width + x] = a; }
		else out[slot] = valid ? a : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	}
}

// Debug render modes (reference render/renderer.cc:62-111, :258-268): one unjittered sample.
// Modes 3 and 6 read an uninitialised tangent frame in the reference; here it is built.
template <int STACK, bool PRIMS>
__global__ void __launch_bounds__(RL_BLOCK)
k_aov(const DRenderParams P, const DSceneView S, float4* __restrict__ out, unsigned long long* __restrict__ counters)
{
	RL_TEX_PROLOGUE(S);
	RL_MATH_PROLOGUE();
	__shared__ int s_stack[STACK * RL_BLOCK];
	int* stk = s_stack + threadIdx.x;
	const uint32_t numSlots = P.numLocalCells * 64u;
	const uint32_t slot = blockIdx.x * RL_BLOCK + threadIdx.x;
	Counters c; c.rays = c.nodes = c.tris = c.shaded = c.texels = c.samples = c.trips = 0; RL_DIAG_BIND(c);
	bool valid = false;
	uint32_t x = 0, y = 0;
	if (slot < numSlots) {
		const uint32_t p = slot & 63u, cellLocal = slot >> 6;
		const uint32_t cell = P.cellFirst + cellLocal * P.cellStride;
		x = (cell % P.cellsX) * 8u + (p & 7u); y = (cell / P.cellsX) * 8u + (p >> 3);
		valid = x < P.width && y < P.height;
	}
	V3 debugValue = v3s(0.0f);
	if (valid) {
		Rng g; g.s = raylib_rng_begin(P.seed, y * P.width + x, 0);
		V3 o, d; float rayTime;
		CameraRay(P.camera, (float)x / (float)P.width, (float)y / (float)P.height, g, o, d, rayTime);
		c.samples++;
		HitRec h;
		if (Traverse<STACK, false, PRIMS>(S, o, d, rayTime, P.rayTMin, h, stk, c)) {
			Surf s;
			const Mat m = LoadMat(S, BuildSurface<PRIMS>(S, o, d, h, s, true, c));
			const uint32_t mode = P.renderMode;
			if (mode == RAYLIB_RENDERMODE_Albedo) {
				debugValue = GetAlbedo(S, m, s.U, s.V, c);
				if (IsMirrorLike(S, m, s.U, s.V, c)) {
					HitRec h2;
					const V3 d2 = reflect(d, s.n);
					if (Traverse<STACK, false, PRIMS>(S, s.p, d2, rayTime, P.rayTMin, h2, stk, c)) {
						Surf s2;
						const Mat m2 = LoadMat(S, BuildSurface<PRIMS>(S, s.p, d2, h2, s2, false, c));
						debugValue = GetAlbedo(S, m2, s2.U, s2.V, c);
					}
				}
			} else if (mode == RAYLIB_RENDERMODE_SurfaceNormal) {
				debugValue = v3s(0.5f) + 0.5f * s.n;
			} else if (mode == RAYLIB_RENDERMODE_MicrosurfaceNormal) {
				V3 N = GetMicrosurfaceNormal(S, m, s, c);
				N = LocalToWorld(s, N);
				debugValue = 0.5f * N + 0.5f;
			} else if (mode == RAYLIB_RENDERMODE_Texcoord) {
				debugValue = v3(s.U, s.V, 0.0f);
			} else if (mode == RAYLIB_RENDERMODE_Emission) {
				debugValue = Emitted(S, m, s, c);
			} else if (mode == RAYLIB_RENDERMODE_Reflectance) {
				V3 refl = v3(1.0f, 0.75f, 0.8f), outD; float pdf, sp;
				Scatter(S, m, d, s, g, c, refl, outD, pdf, sp);
				debugValue = refl;
			}
		}
	}
	if (slot < numSlots) {
		const float4 px = make_float4(debugValue.x, debugValue.y, debugValue.z, 1.0f);
		if (P.rowMajorOutput) { if (valid) out[(size_t)y * P.width + x] = px; }
		else out[slot] = valid ? px : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	}
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t vals[CNT_COUNT] = { c.rays, c.nodes, c.tris, c.shaded, c.texels, c.samples, c.trips };
	for (int k = 0; k < CNT_COUNT; ++k) {
		unsigned long long v = vals[k];
		for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
		if (lane == 0 && v) atomicAdd(&counters[k], v);
	}
}

struct DHitOut { int32_t hit; float t; float p[3]; float n[3]; float paramU, paramV; int32_t material; };

template <int STACK, bool PRIMS>
__global__ void __launch_bounds__(RL_BLOCK)
k_closest_hit(const DSceneView S, const float* __restrict__ rays, int n, float tMin, DHitOut* __restrict__ out)
{
	RL_TEX_PROLOGUE(S);
	RL_MATH_PROLOGUE();
	__shared__ int s_stack[STACK * RL_BLOCK];
	int* stk = s_stack + threadIdx.x;
	const int i = blockIdx.x * RL_BLOCK + threadIdx.x;
	if (i >= n) return;
	Counters c; c.rays = c.nodes = c.tris = c.shaded = c.texels = c.samples = c.trips = 0; RL_DIAG_BIND(c);
	const V3 o = ld3(rays + 6 * i), d = ld3(rays + 6 * i + 3);
	HitRec h;
	DHitOut r; memset(&r, 0, sizeof(r)); r.material = -1;
	if (Traverse<STACK, false, PRIMS>(S, o, d, 0.0f, tMin, h, stk, c)) {
		Surf s;
		const int material = BuildSurface<PRIMS>(S, o, d, h, s, false, c);
		r.hit = 1; r.t = s.t;
		r.p[0] = s.p.x; r.p[1] = s.p.y; r.p[2] = s.p.z;
		r.n[0] = s.n.x; r.n[1] = s.n.y; r.n[2] = s.n.z;
		r.paramU = s.U; r.paramV = s.V; r.material = material;
	}
	out[i] = r;
}

// Image2D::PostProcess, pass 1: maxWhiteLuminance = max(1, max_i luminance_i) (reference render/image.cc:62-72).
// Luminances <= 1 cannot change the result, and for floats >= 1 the ordering of the values is the ordering of
// their bit patterns, so one integer atomicMax per wave suffices; NaN compares false in the reference and is skipped.
__global__ void __launch_bounds__(RL_BLOCK)
k_pp_max(const float4* __restrict__ px, size_t n, unsigned int* __restrict__ whiteBits)
{
	float m = 1.0f;
	for (size_t i = (size_t)blockIdx.x * RL_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * RL_BLOCK) {
		const float4 p = px[i];
		const float L = dot(v3(p.x, p.y, p.z), v3(0.2126f, 0.7152f, 0.0722f));
		if (m < L) m = L;
	}
	for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_down(m, off); if (m < o) m = o; }
	if ((threadIdx.x & 63) == 0 && m > 1.0f) atomicMax(whiteBits, __float_as_uint(m));
}

// pass 2 (reference render/image.cc:76-102)
__global__ void __launch_bounds__(RL_BLOCK)
k_pp_map(float4* __restrict__ px, size_t n, const unsigned int* __restrict__ whiteBits)
{
	RL_MATH_PROLOGUE();
	const size_t i = (size_t)blockIdx.x * RL_BLOCK + threadIdx.x;
	if (i >= n) return;
	const float maxWhiteLuminance = __uint_as_float(*whiteBits);
	float4 p = px[i];
	V3 rgb = v3(p.x, p.y, p.z);
	const float luminanceOld = dot(rgb, v3(0.2126f, 0.7152f, 0.0722f));
	if (luminanceOld <= 0.0001f) rgb = v3s(0.0f);
	else {
		const float numerator = luminanceOld * (1.0f + (luminanceOld / (maxWhiteLuminance * maxWhiteLuminance)));
		const float luminanceNew = numerator / (1.0f + luminanceOld);
		rgb = rgb * (luminanceNew / luminanceOld);
	}
	// min(vec3(1), rgb) with std::min's operand order (core/vec3.h:151-156): (rgb < 1) ? rgb : 1
	rgb = v3(rgb.x < 1.0f ? rgb.x : 1.0f, rgb.y < 1.0f ? rgb.y : 1.0f, rgb.z < 1.0f ? rgb.z : 1.0f);
	const float K = 1.0f / 2.2f;
	p.x = rtm::pow_(rgb.x, K); p.y = rtm::pow_(rgb.y, K); p.z = rtm::pow_(rgb.z, K);
	px[i] = p;
}

// Test hooks: single functions of the hot path evaluated on arrays, so that tests can pin them one by one against
// the reference's outputs (Material::Scatter / ScatteringPdf / Emitted, Camera::GetCameraRay, Texture2D::Sample).
// Record layouts are those of oracle/ref_glue.cc (ref_scatter, ref_camera_rays, ref_texture_sample).
__global__ void __launch_bounds__(RL_BLOCK)
k_eval_scatter(const DSceneView S, int material, const float* __restrict__ in, int n, unsigned long long seed, float* __restrict__ out)
{
	RL_TEX_PROLOGUE(S);
	RL_MATH_PROLOGUE();
	const int i = blockIdx.x * RL_BLOCK + threadIdx.x;
	if (i >= n) return;
	Counters c; c.rays = c.nodes = c.tris = c.shaded = c.texels = c.samples = c.trips = 0; RL_DIAG_BIND(c);
	const float* a = in + 16 * i;
	const V3 d = ld3(a + 3);
	Surf s;
	s.t = a[7]; s.p = ld3(a + 8); s.n = ld3(a + 11); s.U = a[14]; s.V = a[15];
	{
		V3 T = (fabsf(s.n.x) > 0.9f) ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
		V3 B = normalize(cross(T, s.n));
		T = normalize(cross(s.n, B));
		s.tangent = T; s.bitangent = B;
	}
	const Mat m = LoadMat(S, material);
	Rng g; g.s = raylib_rng_begin(seed, (uint32_t)i, 0);
	V3 refl = v3s(0.0f), outD = v3s(0.0f);
	float pdf = 0.0f, sp = 0.0f;
	const bool b = Scatter(S, m, d, s, g, c, refl, outD, pdf, sp);
	const V3 e = Emitted(S, m, s, c);
	float* o = out + 16 * i;
	o[0] = b ? 1.0f : 0.0f;
	o[1] = refl.x; o[2] = refl.y; o[3] = refl.z;
	// the reference leaves the scattered ray default-constructed (0) when a material does not fill it
	const bool wrote = (m.type != MAT_DIFFUSE_LIGHT);
	o[4] = wrote ? outD.x : 0.0f; o[5] = wrote ? outD.y : 0.0f; o[6] = wrote ? outD.z : 0.0f;
	o[7] = wrote ? s.p.x : 0.0f; o[8] = wrote ? s.p.y : 0.0f; o[9] = wrote ? s.p.z : 0.0f;
	o[10] = pdf; o[11] = b ? sp : 0.0f;
	o[12] = e.x; o[13] = e.y; o[14] = e.z;
	o[15] = (m.type == MAT_LAMBERTIAN || m.type == MAT_METAL || m.type == MAT_MICROFACET) ? 2.0f : (m.type == MAT_DIELECTRIC ? 1.0f : 0.0f);
}

__global__ void __launch_bounds__(RL_BLOCK)
k_eval_camera(const DCamera cam, const float* __restrict__ uv, int n, unsigned long long seed, float* __restrict__ out)
{
	RL_MATH_PROLOGUE();
	const int i = blockIdx.x * RL_BLOCK + threadIdx.x;
	if (i >= n) return;
	Rng g; g.s = raylib_rng_begin(seed, (uint32_t)i, 0);
	V3 o, d; float t;
	CameraRay(cam, uv[2 * i], uv[2 * i + 1], g, o, d, t);
	float* r = out + 7 * i;
	r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z; r[6] = t;
}

__global__ void __launch_bounds__(RL_BLOCK)
k_eval_texture(const DSceneView S, int tex, int srgb, const float* __restrict__ uv, int n, float* __restrict__ out)
{
	RL_MATH_PROLOGUE();
	const int i = blockIdx.x * RL_BLOCK + threadIdx.x;
	if (i >= n) return;
	const float4 p = TexFetch(S.textures, S.texels, tex, srgb != 0, uv[2 * i], uv[2 * i + 1]);
	out[4 * i] = p.x; out[4 * i + 1] = p.y; out[4 * i + 2] = p.z; out[4 * i + 3] = p.w;
}

// Test hook: evaluate one device math routine on an array (tests compare with the host libm bit for bit).
__global__ void __launch_bounds__(RL_BLOCK)
k_eval_math(int fn, const float* __restrict__ x, const float* __restrict__ y, int n, float* __restrict__ out)
{
	RL_MATH_PROLOGUE();
	const int i = blockIdx.x * RL_BLOCK + threadIdx.x;
	if (i >= n) return;
	const float a = x[i], b = y ? y[i] : 0.0f;
	float r = 0.0f, s, c;
	switch (fn) {
		case 0: r = rtm::sin_(a); break;
		case 1: r = rtm::cos_(a); break;
		case 2: r = rtm::tan_(a); break;
		case 3: r = rtm::acos_(a); break;
		case 4: r = rtm::asin_(a); break;
		case 5: r = rtm::atan2_(a, b); break;
		case 6: r = rtm::exp_(a); break;
		case 7: r = rtm::log_(a); break;
		case 8: r = rtm::pow_(a, b); break;
		case 9: rtm::sincos_(a, &s, &c); r = s; break;
		case 10: rtm::sincos_(a, &s, &c); r = c; break;
		case 11: r = sqrtf(a); break;
		case 12: r = a / b; break;
		case 13: r = rtm::fmod1_(a); break;
		case 14: r = rtm::rcp1_(a); break;
		case 15: r = rtm::sqrt_(a); break;
		default: break;
	}
	out[i] = r;
}

// Test hook: the short exact sequences against the compiler's IEEE expansions, inside the product library.  out[0] = mismatching cases, out[1] = the
// smallest bit pattern of the swept operand with a mismatch.
//   which 0 / 1: rtm::rcp1_ / rtm::sqrt_ (rl_glibc_math.h) against 1.0f / x and sqrtf(x) on EVERY float bit pattern;
//   which 2: rtm::div_by_(a, b, RN(1 / b)) against a / b -- every bit pattern as the numerator of a set of divisors, and as the divisor of a set of
//            numerators, wherever div_by_'s stated conditions hold (rl_math.h; all significand PAIRS are tools/verify_fastdiv.hip's sweep: this one walks the
//            exponents, the signs and the edges of the conditions);
//   which 3: Barycentric() in its short form against the two divisions and the reference's test -- every bit pattern as X, as Y and as denom of a set of
//            (X, Y, denom) triples, with rden as the host makes it: same verdict, and the same two quotients bit for bit when inside.
__device__ __forceinline__ bool SameBits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__device__ __forceinline__ bool DivByHolds(float a, float b)
{
	const float ma = fabsf(a), mb = fabsf(b);
	if (!(mb >= 0x1p-126f && mb <= 0x1p126f)) return false;
	if (!(ma >= 0x1p-102f && ma <= FLT_MAX)) return false;
	const float q = fabsf(a / b);
	return q >= 0x1p-126f && q < 0x1p127f;
}
__device__ __forceinline__ bool BaryDiffers(float X, float Y, float denom)
{
	const float mag = fabsf(denom);
	float rden;
	if (denom == 0.0f || denom != denom) rden = __uint_as_float(0x7fc00000u);
	else if (mag >= 0x1p-62f && mag <= 0x1p125f) rden = 1.0f / denom;
	else return false;   // such a divisor clears DSceneView::fastBary: the whole scene takes the divisions
	float fa, fb, ea, eb;
	const bool f = Barycentric(true, X, Y, denom, rden, fa, fb), e = Barycentric(false, X, Y, denom, rden, ea, eb);
	return f != e || (f && !(SameBits(fa, ea) && SameBits(fb, eb)));
}
__global__ void __launch_bounds__(RL_BLOCK)
k_verify_exact_math(int which, unsigned long long* __restrict__ out)
{
	const unsigned long long tid = (unsigned long long)blockIdx.x * RL_BLOCK + threadIdx.x, n = (unsigned long long)gridDim.x * RL_BLOCK;
	unsigned long long bad = 0, first = ~0ull;
	// operands the sweeps pair every bit pattern with: ordinary values, the launch's and the triangles' kinds of constants, and the edges of the conditions
	const float fixedB[12] = { 3.0f, 1920.0f, 1080.0f, 0.1f, -7.0f, 3.14159274f, 9.5e10f, 2.4e-9f, 0x1.8p-62f, 0x1.fffffep125f, 0x1p-126f, 0x1p126f };
	const float fixedA[10] = { 1.0f, -3.3f, 1e-20f, 5e20f, 0x1p-102f, 0x1.fffffep-103f, 0x1.234568p-100f, 0.75f, 1919.0f, 0x1.fffffep127f };
	for (unsigned long long b = tid; b < (1ull << 32); b += n) {
		const float x = __uint_as_float((uint32_t)b);
		bool differs = false;
		if (which <= 1) {
			const float want = which == 0 ? 1.0f / x : __builtin_sqrtf(x);
			const float got = which == 0 ? rtm::rcp1_(x) : rtm::sqrt_(x);
			differs = !SameBits(want, got);   // a NaN must meet a NaN
		} else if (which == 2) {
			for (int k = 0; k < 12; ++k) { const float d = fixedB[k]; if (DivByHolds(x, d) && !SameBits(rtm::div_by_(x, d, 1.0f / d), x / d)) differs = true; }
			for (int k = 0; k < 10; ++k) { const float a = fixedA[k]; if (DivByHolds(a, x) && !SameBits(rtm::div_by_(a, x, 1.0f / x), a / x)) differs = true; }
		} else {
			// (X, Y, denom) of ordinary hits, of hits on an edge and at a vertex, of misses by a hair, with tiny, huge and special members
			const float T[14][3] = { { 1.0e9f, 2.0e9f, 9.5e10f }, { -1.0e9f, -2.0e9f, -9.5e10f }, { 0.0f, 4.0e10f, 9.5e10f }, { -0.0f, 0.0f, 9.5e10f }, { 1e-3f, 9.4999e10f, 9.5e10f },
			                         { 3e-12f, 1.0f, 2.4e-9f }, { -3e-12f, 1.0e-10f, 2.4e-9f }, { 1e-30f, 1e-31f, 0x1p-62f }, { 5e-20f, 2e-21f, 0x1.8p-60f }, { 1e-42f, 1e-10f, 1e-9f },
			                         { 0x1p100f, 0x1p99f, 0x1p125f }, { 0x1.fffffep127f, 1.0f, 2.0f }, { 4.75e10f, 4.75e10f, 9.5e10f }, { 4.7500004e10f, 4.75e10f, 9.5e10f } };
			for (int k = 0; k < 14; ++k) {
				differs = differs || BaryDiffers(x, T[k][1], T[k][2]) || BaryDiffers(T[k][0], x, T[k][2]) || BaryDiffers(T[k][0], T[k][1], x);
				differs = differs || BaryDiffers(x, -T[k][1], -T[k][2]) || BaryDiffers(x, x, T[k][2]);
			}
		}
		if (differs) { ++bad; first = min(first, b); }
	}
	if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], first); }
}

// Raylib_DumpImageData's packing (reference render/image.cc:121-135: RGB, 12 bytes per pixel, row-major) on the device: 25 MB cross the bus instead of 33,
// and the host is left with a plain copy.  One thread packs four pixels: four 16-byte loads, three 16-byte stores.
__global__ void __launch_bounds__(RL_BLOCK)
k_pack_rgb(const float4* __restrict__ px, float4* __restrict__ out, float* __restrict__ outTail, size_t n)
{
	const size_t q = (size_t)blockIdx.x * RL_BLOCK + threadIdx.x;
	const size_t i = q * 4;
	if (i + 4 <= n) {
		const float4 a = px[i], b = px[i + 1], c = px[i + 2], d = px[i + 3];
		out[q * 3] = make_float4(a.x, a.y, a.z, b.x); out[q * 3 + 1] = make_float4(b.y, b.z, c.x, c.y); out[q * 3 + 2] = make_float4(c.z, d.x, d.y, d.z);
	} else {
		for (size_t k = i; k < n; ++k) { const float4 a = px[k]; outTail[3 * k] = a.x; outTail[3 * k + 1] = a.y; outTail[3 * k + 2] = a.z; }
	}
}

// The frame from the ranks' cell buffers (N > 1 behind Raylib_Render): cell c was rendered by rank c % N as its (c / N)-th cell.
struct ScatterPlan { uint32_t ranks; uint32_t offset[16]; };   // offset[r]: first float4 of rank r's cells in the gather buffer
__global__ void __launch_bounds__(RL_BLOCK)
k_scatter_cells(const float4* __restrict__ gather, float4* __restrict__ out, uint32_t width, uint32_t height, uint32_t cellsX, const ScatterPlan plan)
{
	const size_t i = (size_t)blockIdx.x * RL_BLOCK + threadIdx.x;
	if (i >= (size_t)width * height) return;
	const uint32_t x = (uint32_t)(i % width), y = (uint32_t)(i / width);
	const uint32_t cell = (y >> 3) * cellsX + (x >> 3);
	const uint32_t rank = cell % plan.ranks, local = cell / plan.ranks;
	out[i] = gather[plan.offset[rank] + local * 64u + ((y & 7u) << 3) + (x & 7u)];
}

#endif   // RL_TU_POOL

} // namespace rl

#ifndef RL_TU_POOL
#include "rl_runtime.inl"
#endif
