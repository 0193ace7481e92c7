// Device-side data layout (HBM) shared by the host flattener and the HIP kernels.
// Every record is a multiple of 16 bytes and 64-byte records are 64-byte aligned,
// so one lane fetches a record with 4 x global_load_dwordx4 and never straddles a
// 128-byte line more than once.
#pragma once

#include <stdint.h>

namespace rl {

// BVH2 node, 64 B: both child boxes live in the parent, so one fetch decides both
// children (the reference keeps one box per heap node and chases a pointer per
// child: geom/bvh.h:20-22, 48 B + vtable).
//   child >= 0            : inner node index
//   child <  0, != EMPTY  : leaf, ~child = (first << 6) | (kind << 4) | (alphaTested << 3) | (count - 1)
//                           kind 0: `count` (<= 4) triangles from `first`; kind 1 / 2: sphere / cube number `first`
//   child == DNODE_EMPTY  : nothing (box is inverted, never hit)
#define DNODE_EMPTY ((int32_t)0x80000000)
struct alignas(64) DNode {
	float lmin[3], lmax[3];
	float rmin[3], rmax[3];
	int32_t left, right;
	int32_t pad0, pad1;
};
static_assert(sizeof(DNode) == 64, "DNode");

// BVH4 node, 128 B = one L2 line: up to four child boxes, structure of arrays so that a lane reads
// lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] child[4] with 7 x 16-byte loads.  Child references as in DNode.
// Built by collapsing the BVH2 (rl_bvh.cc); used by the pool schedule on large scenes, where traversal is bound by
// dependent fetches: half as many of them per ray.
struct alignas(128) DNode4 {
	float lo[3][4];
	float hi[3][4];
	int32_t child[4];
	uint32_t pad[4];
};
static_assert(sizeof(DNode4) == 128, "DNode4");

// The same node on an 8-bit grid, 64 B (half an L2 line; large scenes wait on node data, and this halves the lines a step touches).
// Child k's box along axis a is [origin[a] + qlo_a,k * step_a, origin[a] + qhi_a,k * step_a], step_a = 2^(e_a - 127): the grid is
// anchored at the minimum corner of the node's own box, lower planes are rounded down and upper planes up, so a child's grid box
// CONTAINS its float box.  The closest hit does not depend on how tight a box is (candidate rule, rl_render.hip OwnBoxPass): the
// image stays bit-identical.  stepX / stepY / stepZ: the grid steps as floats (powers of two), in the fourth word of the three 16-byte rows (round 3:
// one word of packed exponents before -- three shifts and three masks per node step to unpack, on a kernel that is bound by VALU issue).
// qlo[a] / qhi[a]: byte k = child k.  An unused child has child[k] == DNODE_EMPTY (and an inverted grid box).
struct alignas(64) DNode4Q {
	float origin[3]; float stepX;
	uint32_t qlo[3]; float stepY;
	uint32_t qhi[3]; float stepZ;
	int32_t child[4];
};
static_assert(sizeof(DNode4Q) == 64, "DNode4Q");
// The 8-wide grid node, 80 bytes = five 16-byte loads (round 4).  The tree walk of a large scene is a chain of dependent steps -- fetch a node (about a
// microsecond under load), test its boxes, fetch the next -- of which a SIMD keeps four in flight: fewer loads per step, fuller turns of
// the vote and cheaper pops did not shorten it; FEWER STEPS do.  Eight children per node: 0.6 x the steps of the 4-wide tree.
//   * boxes as in DNode4Q: the node's minimum corner as three floats, a power-of-two step per axis (here as biased exponents, one byte each), 8-bit planes
//     (byte c of qlo[a][c >> 2] / qhi[a][c >> 2] = child c); an unused child has the inverted box lo 255 / hi 0, which no ray enters;
//   * children are NOT named one by one: the inner children of a node are consecutive nodes (childBase + rank among the inner children, in slot order:
//     imask bit c = child c is an inner node), the triangles of its leaf children are consecutive triangle slots (triBase + rank among the bits of leafMask,
//     whose nibble c holds as many low bits as leaf child c has triangles, <= 4) -- so a traversal keeps the hit children of a node as ONE stack entry
//     (base, hit bits) however many they are, and needs no sort: children sit in the slot whose three bits say on which side of the node they lie
//     (bit 0: +x, bit 1: +y, bit 2: +z; the builder assigns them greedily), and a ray visits the slots in the order slot XOR (signs of its direction)
//     (Ylitie, Karras, Laine: "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", HPG 2017 -- the layout idea; the arithmetic here
//     is this library's: exact-corner grid boxes, the candidate rule, the tie rule);
//   * alphaMask bit c = leaf child c holds a triangle whose material is alpha-tested (the cut-out test runs inside traversal, DNode's flag).
// Round 5 built and measured the other end of the trade (git 530c313: a 128-byte node of half-float planes, one v_fma_mix_f32 per plane -- 4 issue clocks where a byte
// costs v_cvt_f32_ubyte + v_fma_f32 = 6 -- near / far rows by XOR, the ray's own factors: 169 VALU instructions per step where this node takes 248): the vector memory
// path charges a wave one clock per lane and LOAD INSTRUCTION whatever its width (tools/vmem_width_bench.hip: 0.88 clocks, dword or dwordx4; four lanes of a quad
// on one address count once), eight loads per step put the walk on that path's ceiling (TCP_TOTAL_CACHE_ACCESSES 1.05 per clock and CU of 1.14) as the 248
// instructions had it on the VALU port's, and the frames came out - 2 % (298 k triangles), + 8 % (2.36 M), + 17 % (10.1 M).  Five loads and bytes stay.
struct alignas(16) DNode8 {
	float origin[3];
	uint32_t meta;        // ex | ey << 8 | ez << 16 (biased exponents of the steps) | imask << 24
	uint32_t childBase;   // node index of the first inner child
	uint32_t triBase;     // triangle slot of the first triangle of the first leaf child
	uint32_t leafMask;    // nibble c: (1 << count) - 1 for leaf child c, else 0
	uint32_t alphaMask;   // bit c (c < 8)
	uint32_t qlo[3][2], qhi[3][2];
};
static_assert(sizeof(DNode8) == 80, "DNode8");
#define RL_POOL8_MAXLEVELS 16    /* levels of an 8-wide tree the pool kernel's stack of groups holds (k_trace_pool<2 * this, ..., 3>; rl_runtime.inl selects the walk only then) */
// An LDS copy of the top of the 8-wide tree (round 5, measured and not kept: RL_TOP8_NODES > 0 builds it).  Nodes are numbered breadth first, so the first N of them
// are the levels every ray starts with -- 23 % of all node steps on the 298 k-triangle room for N = 73, 10 % on the 10 M-triangle one (diagnostic build
// -DRL_DIAG_TOPN) -- and a step on one of them costs the vector memory path nothing (ds_read_b128 instead of global_load_dwordx4) and waits for no cache.  Four
// workgroups per CU leave room for it only if the LDS part of the stack shrinks from eight groups to four (RL_POOL8_LSTACK 8: 113 nodes of 80 bytes); the 298 k
// room never holds more than four groups in 99.6 % of its steps, the 10 M room in 94.5 % -- and there a pop from the private overflow is one more trip to memory
// in the chain.  With this node (the walk sits on the VALU port, not on memory): 298 k room 90.4 ms with and without the copy, 2.36 M + 2.7 %, 10.1 M + 7.8 %.
// (With the 128-byte half-float node, which sits on the vector memory path: 93.9 -> 90.1 ms.)
#ifndef RL_TOP8_NODES
#define RL_TOP8_NODES 0
#endif
#ifndef RL_POOL8_LSTACK
#define RL_POOL8_LSTACK 16       /* words of the 8-wide walk's stack that live in LDS: two per group */
#endif
// which of the two the POOL schedule walks (a build-time switch so that both can be timed: make variant EXTRA=-DRL_Q4=0);
// k_trace has both as instantiations and takes the float boxes whenever the scene carries them
#ifndef RL_Q4
#define RL_Q4 1
#endif

// Triangle intersection record, 64 B.  The reference tests ray vs plane, then
// barycentrics from dot products of edge vectors (geom/triangle.cc:18-58); all
// ray-independent terms of that formula are precomputed here with the reference's
// own operation order, so the per-ray arithmetic reproduces its t / barycentrics
// bit for bit:  u = v1-v0, v = v2-v0, uv = dot(u,v), uu, vv, denom = uv*uv - uu*vv.
// denom itself is three operations on values the record holds (the device forms it the way the host did); the sixteenth float is its correctly
// rounded reciprocal, with which the two divisions by denom become six issue cycles each (rl_math.h div_by_, rl_render.hip Barycentric).
struct alignas(64) DTriIsect {
	float v0[3];
	float n[3];      // unit geometric normal, normalize(cross(v1-v0, v2-v0)) (geom/triangle.h:34-38)
	float v1[3];     // the other two vertices: the edges u = v1 - v0, v = v2 - v0 are formed on the device (the reference's own
	float v2[3];     // subtraction), and the exact AABB of the three vertices is what the candidate rule tests the ray against
	float uv, uu, vv;
	float rden;      // RN(1 / denom) for 2^-62 <= |denom| <= 2^125; NaN for denom == 0 or NaN (never hit either way); any other divisor clears DSceneView::fastBary
};
static_assert(sizeof(DTriIsect) == 64, "DTriIsect");

// Triangle shading record, 64 B: fetched once per path vertex for the winning hit
// (and for alpha-tested candidates).
struct alignas(64) DTriShade {
	float n0[3], n1[3], n2[3];
	float s0, t0, s1, t1, s2, t2;
	int32_t material;
};
static_assert(sizeof(DTriShade) == 64, "DTriShade");

// Analytic primitives of the reference's procedural scenes (geom/sphere.h:22-25, geom/cube.h:36-41).
struct alignas(16) DSphere { float center[3]; float radius; int32_t material; int32_t pad[3]; };
static_assert(sizeof(DSphere) == 32, "DSphere");
struct alignas(16) DCube { float minBounds[3]; float timeStartMove; float maxBounds[3]; int32_t material; float velocity[3]; int32_t pad; };
static_assert(sizeof(DCube) == 48, "DCube");

// Material record, 80 B (fields as reference render/material.h, see rl_host.h HostMaterial).
struct alignas(16) DMaterial {
	int32_t type;
	float albedo[3];
	float roughness, metallic;
	float emissive[3];
	float ior;
	float transmission[3];
	float fuzziness;
	int32_t tex[5];
	int32_t pad;
};
static_assert(sizeof(DMaterial) == 80, "DMaterial");

// Texture descriptor; texels are float RGBA (16 B) in one pool, row 0 = top
// (reference render/image.h:88-119 keeps float Pixels too).
#define RL_LDS_TEXTURES 56   /* texture descriptors a workgroup keeps in LDS (896 B: what the pool kernel's 40 KB share of a CU's LDS has left) */
struct DTexture {
	uint32_t offset;   // first texel in the pool
	int32_t width, height;
	int32_t pad;
};

// Camera constants (reference render/camera.h:95-101).
struct DCamera {
	float origin[3];     float lensRadius;
	float top_left[3];   float beginTime;
	float horizontal[3]; float timePeriod;
	float vertical[3];   float pad0;
	float u[3];          float pad1;
	float v[3];          float pad2;
};

// Scenes of at most 4 * RL_LEAFLIST_RECORDS leaves carry a leaf list (rl_bvh.cc; k_trace<..., LDS = 2> walks it instead of the tree).
#ifndef RL_LEAFLIST_RECORDS
#define RL_LEAFLIST_RECORDS 6
#endif
// ... and at most this many triangles: 4.5 per leaf on average -- beyond, the leaves grow towards 8 triangles and the tree wins again (tools/gpu_leaflist.py)
#define RL_LEAFLIST_MAXTRIS (RL_LEAFLIST_RECORDS * 18)

struct DSceneView {
	const DNode* nodes;
	const DNode4Q* nodes4;     // the wide tree on the 8-bit grid: what the pool schedule walks (nullptr: the scene has none)
	const DNode4* nodes4f;     // the wide tree with float boxes: uploaded for small, cache-resident scenes (k_trace), where the grid saves nothing
	const DNode8* nodes8;      // the 8-wide tree (the pool schedule's default when the scene carries one; nullptr: none)
	int32_t numNodes8;
	const DTriIsect* isect;
	const DTriShade* shade;
	const int32_t* alphaTex;   // per triangle slot: the texture a cut-out test of that triangle reads (its material's converted albedo map), -1: none.  Lets the test inside the walk
	                           // start its texel fetch from the triangle alone instead of triangle -> material -> texture (nullptr: no material of the scene has a map)
	const DMaterial* materials;
	const DTexture* textures;
	int32_t numTextures;       // (incl. the converted copies of albedo maps) up to RL_LDS_TEXTURES descriptors are read from a copy in LDS: rl_render.hip TexTable
	const float* texels;       // float4 pool
	const DSphere* spheres;
	const DCube* cubes;
	float sunIlluminance[3];
	float sunDirection[3];     // normalised
	const float* sky;          // (float4 per texel) the panorama's texels as of the start of this render (row 0 = top); nullptr = none
	int32_t skyWidth, skyHeight;
	int32_t hasSun;
	int32_t numTriangles;
	int32_t numNodes4, numMaterials;   // for the LDS-resident copy of a small scene (k_trace<..., LDS>)
	const DNode4* leafList;    // scenes of few leaves: the leaves' boxes, four to a record (k_trace<..., LDS = 2>); nullptr = none
	int32_t numLeafRecords;
	int32_t fastBary;          // every triangle's rden is usable (above): the barycentric divisions take the short form
};

// Counters written by the kernels (one 64-bit atomic per wave and counter at exit).
enum { CNT_RAYS = 0, CNT_NODES, CNT_TRIS, CNT_SHADED, CNT_TEXELS, CNT_SAMPLES, CNT_TRIPS, CNT_COUNT };

struct DRenderParams {
	uint32_t width, height;
	uint32_t spp;              // max(1, samplesPerPixel): divisor of the mean
	uint32_t sampleBegin;      // first sample index of this batch
	uint32_t sampleCount;      // samples in this batch
	int32_t  maxPathLength;
	float    rayTMin;
	uint32_t renderMode;
	uint64_t seed;
	uint32_t cellsX, cellsY;
	uint32_t cellFirst, cellStride, numLocalCells;
	uint32_t numJobs;          // numLocalCells * sampleCount * 64
	uint32_t stackStride;      // threads in the grid (path-stack column count)
	uint32_t rowMajorOutput;   // 1: out[y*W+x]; 0: out[localCell*64 + p]
	uint32_t jobChunk;         // jobs a wave takes from a head of the job list per atomic
	uint32_t numHeads;         // heads of the job list: 8 (one per XCD) or 1; head h covers jobs [h, h + 1) * jobsPerHead (rl_render.hip TakeJobs)
	uint32_t jobsPerHead;      // whole cells: a multiple of 64 * sampleCount
	uint32_t guideShift;       // 0: every draw asks for jobChunk jobs; s > 0: at most (jobs left in the band at the wave's previous draw) >> s (TakeJobs)
	uint32_t padQueue;
	float    invWidth, invHeight;   // RN(1 / (float)width), RN(1 / (float)height): the pixel -> [0, 1) divisions of GenerateCell (rl_render.hip PixelUV)
	// Cells no camera ray of which can meet the scene's bounding box (host, rl_runtime.inl CullCells: pinhole camera, no sky panorama, the frame's box on the
	// image plane with a margin) are not in the job list: activeCells[i] is the i-th listed local cell (nullptr: all numLocalCells cells, in order), and
	// k_resolve sums the constant every one of their samples would have come to -- emptyL, the sun's illuminance or nothing -- for cells flagged in cellEmpty.
	const uint32_t* activeCells;
	const uint8_t*  cellEmpty;
	uint32_t numActiveCells;
	float    emptyL[3];        // the sun's part of that constant (0 + illuminance, or 0)
	uint32_t emptySky;         // 1: the scene has a sky panorama -- a dropped cell's samples differ by their sky texel: k_resolve generates each sample's camera ray and looks it up
	uint32_t magicSamples;     // floor(2^32 / sampleCount), floor(2^32 / cellsX): division by multiply-high in DecodeJob
	uint32_t magicCellsX;
	uint64_t seedMixed;        // raylib_rng_mix64(seed), hoisted out of the per-sample stream set-up
	DCamera camera;
};

} // namespace rl
