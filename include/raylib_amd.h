/*
 * raylib_amd.h -- ADDITIONAL exports of the MI355X raylib.  Nothing here exists in
 * the reference; raylib.h alone is the drop-in surface.  These exist because the
 * reference has no seed, no counters and no device boundary:
 *   - a deterministic seed (RendererSettings has no seed field and
 *     must not grow one, reference raylib_types.h:41-57)
 *   - ray / node / triangle counters (the reference cannot report Mrays/s,
 *     render/renderer.cc:114-208 has no counter)
 *   - a render entry that leaves the pixels in HBM and can restrict the work to a
 *     strided subset of the 8x8 cells (reference render/renderer.cc:21-22,305-319),
 *     which is how bench.py tiles an image over N ranks (one process per GPU)
 *   - host-logic introspection for CPU-only tests (flattened scene, BVH)
 *
 * Environment variables read by the library:
 *   RAYLIB_SEED    default seed (decimal, default 1) when RaylibAMD_SetSeed was not called
 *   RAYLIB_DEVICE  HIP device ordinal to use (default: LOCAL_RANK if set, else 0)
 *   RAYLIB_NUM_GPUS  N: Raylib_Render (and every whole-frame render) splits the frame's 8x8 cells round-robin over N devices of
 *                  this process -- devices RAYLIB_DEVICE .. RAYLIB_DEVICE + N - 1, or the list RAYLIB_GPU_MAP="d0,d1,..." (one
 *                  entry per rank; a device may be named more than once, which puts several ranks on it: tests) -- and gathers
 *                  the cells on the first device.  The frame is bit-identical to the one-device frame.  Default 1.
 *   RAYLIB_GATHER  rccl (default: grouped ncclSend / ncclRecv, librccl loaded at run time) | peer (hipMemcpyPeerAsync pushes)
 */
#ifndef RAYLIB_AMD_H
#define RAYLIB_AMD_H

#include "raylib_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RaylibAMDStats {
	/* Every counter below counts work a KERNEL EXECUTED.  Camera samples of cells that were dropped from the job list (culledCells, below) are not in them. */
	uint64_t rays;            /* closest-hit + occlusion queries executed on the device (reference renderer.cc:129,194,70,79); a camera ray or sun ray that is
	                             decided by the root node's two boxes alone is one query that fetched one node record */
	uint64_t nodesVisited;    /* 64-byte BVH node records fetched (a float-box BVH4 node or a leaf-list record of four boxes counts as two) */
	uint64_t trisTested;      /* 64-byte triangle intersection records fetched */
	uint64_t shadedHits;      /* 64-byte triangle shading records fetched: one per shaded hit and one per cut-out candidate tested during a walk (the latter depends on the walk's order) */
	uint64_t texFetches;      /* 16-byte texels fetched (incl. the sky texel k_resolve looks up per sample of a cell outside the scene's silhouette) */
	uint64_t cameraSamples;   /* (pixel, sample) paths generated and traced by the megakernel (culledSamples are NOT in here: cameraSamples + culledSamples = pixels x spp) */
	uint64_t pixels;          /* pixels written (16 bytes each) */
	double   kernelMs;        /* HIP-event time of all kernels of the last render, on the library's stream */
	double   traceKernelMs;   /* ... of the path-tracing megakernel launches only */
	double   wallMs;          /* host wall clock of the last render call, incl. D2H copy when made */
	uint32_t traceLaunches;   /* megakernel launches in the last render (one per sample batch) */
	uint32_t numNodes;        /* BVH nodes of the scene */
	uint32_t numTriangles;
	uint32_t bvhDepth;
	uint64_t waveTrips;       /* bounce-loop trips summed over waves (a diagnostic: it depends on which wave drew which batch and differs from run to run) */
	uint32_t pathsPerWave;    /* schedule of the megakernel: 64 = k_trace (one path per lane), 128/192/256 = k_trace_pool */
	uint32_t ranks;           /* logical ranks (devices) that rendered the frame: 1, or RAYLIB_NUM_GPUS for a whole-frame render */
	/* ---- a whole-frame render over several ranks (RAYLIB_NUM_GPUS > 1): where the time went, so that a scaling loss can be attributed ---- */
	uint32_t gatherMode;      /* how the ranks' cells reached rank 0's device: 0 nothing to move (one rank, or every rank on rank 0's device),
	                             1 RCCL grouped ncclSend / ncclRecv, 2 hipMemcpyPeerAsync pushes (RAYLIB_GATHER=peer, or RCCL could not be initialised) */
	uint32_t rcclCommSize;    /* devices in the library's RCCL communicator (0: not initialised) */
	uint32_t devices;         /* distinct physical devices the ranks ran on */
	uint32_t jobHeads;        /* heads of the job list in the last megakernel launch: 8 = one per XCD (csrc/rl_render.hip TakeJobs), RAYLIB_JOB_HEADS overrides */
	double   gatherMs;        /* on rank 0's stream: from the end of rank 0's own kernels until every rank's cells are on its device (waiting for slower ranks included) */
	double   scatterMs;       /* k_scatter_cells: cell buffers -> row-major frame */
	double   rankKernelMs[16];/* per rank: HIP-event time of all its kernels (kernelMs is their maximum) */
	double   rankTraceMs[16]; /* per rank: ... of its megakernel launches (traceKernelMs is their maximum) */
	/* ---- cells outside the scene's silhouette (csrc/rl_cull.cc): dropped from the job list, filled with the miss shader's constant by k_resolve.
	 *      Nothing below was executed by any kernel; these are what the dropped samples WOULD have cost (one root-box query each, two with a sun),
	 *      kept apart so that a rate computed from `rays` is a rate of executed queries. ---- */
	uint32_t culledCells;     /* 8 x 8 cells left out of the job list (summed over ranks); RaylibAMD_CullCells on the same view returns this number */
	uint32_t listedCells;     /* cells in the job list (culledCells + listedCells = the frame's cells) */
	uint64_t culledSamples;   /* camera samples of those cells: pixels of the culled cells x spp */
	uint64_t culledRays;      /* queries (= root node records) those samples stand for: culledSamples x (2 with a sun, else 1) */
	/* ---- which tree the megakernel walked ---- */
	uint32_t treeWidth;       /* children per node: 2, 4 (64-byte grid nodes), 8 (80-byte grid nodes, octant-ordered children; scenes whose rays are expected to take
	                             many steps, RAYLIB_BVH8=0|1 overrides) or 0 = no tree (the leaf list of a scene of few leaves) */
	uint32_t nodeBytes;       /* bytes of one record counted in nodesVisited: 64, or 80 for the 8-wide tree */
} RaylibAMDStats;

/* Seed of the per-(pixel, sample) streams of include/raylib_amd_rng.h. */
RAYLIB_API void     RaylibAMD_SetSeed(uint64_t seed);
RAYLIB_API uint64_t RaylibAMD_GetSeed(void);

/* Stats of the last Raylib_Render / RaylibAMD_RenderDevice on this thread's library state. */
RAYLIB_API void RaylibAMD_GetLastStats(RaylibAMDStats* outStats);

/* 1 when a gfx950-capable HIP device is present and the kernels are loadable. */
RAYLIB_API int32_t RaylibAMD_DeviceAvailable(void);
/* 16 hex digits: SHA-256 prefix of the device sources + build flags this library was made from (software-raytracing_amd/Makefile BUILD_ID).
 * Hardware-counter profiles kept under profiles/ carry the id of the library they were taken from; bench.py compares. */
RAYLIB_API const char* RaylibAMD_BuildId(void);

/*
 * Render the cells {cellFirst, cellFirst + cellStride, ...} (8x8-pixel cells numbered
 * row-major over ceil(W/8) x ceil(H/8)) and leave RGBA float pixels in device memory.
 *   outDevice: device pointer; when cellStride == 1 && cellFirst == 0 it receives the
 *              row-major W*H*4-float image; otherwise it receives the rank's cells
 *              back to back, 64 pixels (row-major inside the cell) * 4 floats each.
 *              Must hold RaylibAMD_CellBufferFloats(...) floats.  May be 0: the
 *              library then renders into its own buffer (bench timing without output).
 * Returns 1 on success, 0 on failure (no device, bad handle).  Synchronous: the
 * library's stream has been synchronised when it returns.
 */
RAYLIB_API int32_t RaylibAMD_RenderDevice(const RendererSettings* settings, SceneHandle scene,
	CameraHandle camera, uint32_t cellFirst, uint32_t cellStride, void* outDevice);
/* Same, but the result is copied to host memory (RaylibAMD_CellBufferFloats(...) floats). */
RAYLIB_API int32_t RaylibAMD_RenderCellsHost(const RendererSettings* settings, SceneHandle scene,
	CameraHandle camera, uint32_t cellFirst, uint32_t cellStride, float* outHost);
RAYLIB_API uint64_t RaylibAMD_CellBufferFloats(uint32_t width, uint32_t height, uint32_t cellFirst, uint32_t cellStride);
RAYLIB_API uint32_t RaylibAMD_NumCells(uint32_t width, uint32_t height);

/* Closest-hit queries on the flat BVH (rays: n*6 floats o,d; out: n*11 words
 * {hit, t, p[3], n[3], paramU, paramV, material} as in oracle/flat_scene.h FlatHit). */
RAYLIB_API int32_t RaylibAMD_ClosestHit(SceneHandle scene, const float* rays, int32_t n, float tMin, void* outHits);

/* ---- procedural scene elements ----------------------------------------------------------
 * The reference's two procedural demo scenes (src/main.cc:913-984) `new` its C++ classes (Sphere, Cube, Triangle,
 * Lambertian, Metal, ...) in the application and pass the object pointers to Raylib_AddSceneElement.  A C ABI
 * cannot accept foreign C++ objects, so the same elements are created through the library instead; the handles
 * returned here are what Raylib_AddSceneElement accepts.  Elements and materials are owned by the library and
 * BORROWED by scenes (destroy them after the scene, like OBJ models). */
typedef uintptr_t MaterialHandle;
/* type: 0 Lambertian(albedo) 1 Mirror(albedo) 2 Dielectric(ior, transmission) 3 Microfacet(albedo, roughness, metallic, emissive)
 *       4 Metal(albedo, fuzziness) 5 DiffuseLight(albedo = intensity)      (reference render/material.h:50-270) */
RAYLIB_API MaterialHandle RaylibAMD_CreateMaterial(int32_t type, const float albedo[3], float roughness, float metallic,
	const float emissive[3], float ior, const float transmission[3], float fuzziness);
RAYLIB_API int32_t RaylibAMD_DestroyMaterial(MaterialHandle material);
/* reference geom/sphere.h:11-16 */
RAYLIB_API SceneElementHandle RaylibAMD_CreateSphere(float cx, float cy, float cz, float radius, MaterialHandle material);
/* reference geom/cube.h:24-31 (Cube::FromMinMaxBounds) */
RAYLIB_API SceneElementHandle RaylibAMD_CreateCube(const float minBounds[3], const float maxBounds[3], float timeStartMove,
	const float velocity[3], MaterialHandle material);
/* reference geom/triangle.h:12-15; UVs as SetParameterization (s0 t0 s1 t1 s2 t2), may be NULL */
RAYLIB_API SceneElementHandle RaylibAMD_CreateTriangle(const float v0[3], const float v1[3], const float v2[3],
	const float n0[3], const float n1[3], const float n2[3], const float uv[6], MaterialHandle material);
RAYLIB_API int32_t RaylibAMD_DestroySceneElement(SceneElementHandle element);

/* Test hooks: one function of the hot path on an array of inputs, evaluated by the device code the megakernel uses.
 *   EvalScatter   : Material::Scatter + ScatteringPdf + Emitted for scene material `material`; record i uses the stream
 *                   (seed, i, 0).  in: 16 floats (ray o, d, time; hit t, p, n, paramU, paramV); out: 16 floats (scattered?,
 *                   reflectance, direction, origin, pdf, scatteringPdf, emitted, draws) -- layouts of oracle/ref_glue.cc.
 *   EvalCameraRays: Camera::GetCameraRay(u, v), stream (seed, i, 0); out 7 floats (o, d, time).
 *   EvalTexture   : Texture2D::Sample of scene texture `texture`; out 4 floats. */
RAYLIB_API int32_t RaylibAMD_EvalScatter(SceneHandle scene, int32_t material, const float* records, int32_t n, uint64_t seed, float* out);
RAYLIB_API int32_t RaylibAMD_EvalCameraRays(CameraHandle camera, const float* uv, int32_t n, uint64_t seed, float* out);
RAYLIB_API int32_t RaylibAMD_EvalTexture(SceneHandle scene, int32_t texture, int32_t bSRGB, const float* uv, int32_t n, float* out);
/* Test hook: out[i] = f(x[i] [, y[i]]) evaluated by the DEVICE math the megakernel uses (csrc/rl_math.h).
 * fn: 0 sinf, 1 cosf, 2 tanf, 3 acosf, 4 asinf, 5 atan2f(x,y), 6 expf, 7 logf, 8 powf(x,y), 9/10 sincos (sin / cos
 * part), 11 sqrtf, 12 x / y, 13 fmodf(x, 1).  y may be NULL for one-argument functions. */
RAYLIB_API int32_t RaylibAMD_EvalDeviceMath(int32_t fn, const float* x, const float* y, int32_t n, float* out);

/* Test hook (host only, no device needed): which 8 x 8 cells of a width x height frame can no ray of the camera -- pinhole or thin lens, no sky panorama -- meet the box
 * bounds = { min x, y, z, max x, y, z } in?  The renderer leaves those cells out of the megakernel's job list and fills them with the miss shader's
 * constant (csrc/rl_cull.cc).  outEmpty: one byte per cell, row-major, 1 = dropped; outConstant: the constant (the sun's illuminance, or nothing).
 * The sun direction is the normalised one the scene holds.  Returns the number of dropped cells, 0 when none can be dropped, -1 when the frame is not
 * eligible (a corner of the box beside or behind the camera, a sun ray from the camera -- from any point of its lens -- that may meet the box). */
RAYLIB_API int32_t RaylibAMD_CullCells(CameraHandle camera, const float* bounds, const float* sunIlluminance, const float* sunDirection,
                                       int32_t width, int32_t height, uint8_t* outEmpty, float* outConstant);

/* Test hook: the device's short exact sequences for 1.0f / x (which = 0) and sqrtf(x) (which = 1) -- csrc/rl_glibc_math.h rcp1_ / sqrtf_, used by
 * normalize and every reciprocal of the shading code -- against the compiler's IEEE expansions on ALL 2^32 float bit patterns, on the device.
 * which = 2: a / b with the divisor's correctly rounded reciprocal in hand (csrc/rl_math.h div_by_: the pixel -> [0, 1) divisions of a camera ray and the two
 * barycentric divisions of a triangle test) -- every bit pattern as numerator of a set of divisors and as divisor of a set of numerators, wherever the
 * sequence's stated conditions hold; which = 3: the triangle test's short barycentric form (csrc/rl_render.hip Barycentric) against the two divisions and
 * the reference's test, every bit pattern in each of its three operands: same verdict, same quotients.
 * outMismatches: inputs whose results differ (a NaN may differ in payload); outFirstBits: the smallest such bit pattern.  Returns 1 when the sweep ran. */
RAYLIB_API int32_t RaylibAMD_VerifyExactMath(int32_t which, uint64_t* outMismatches, uint64_t* outFirstBits);

/* ---- host-logic introspection (no GPU needed) ---------------------------------- */
/* Flattened scene as the kernels see it.  Triangle record = 26 words, material record =
 * 19 words, both laid out as oracle/flat_scene.h FlatTriangle / FlatMaterial. */
RAYLIB_API int32_t RaylibAMD_SceneNumTriangles(SceneHandle scene);
RAYLIB_API int32_t RaylibAMD_SceneNumMaterials(SceneHandle scene);
RAYLIB_API int32_t RaylibAMD_SceneNumTextures(SceneHandle scene);
RAYLIB_API void    RaylibAMD_SceneExportTriangles(SceneHandle scene, void* outTriangles);
RAYLIB_API void    RaylibAMD_SceneExportMaterials(SceneHandle scene, void* outMaterials);
RAYLIB_API void    RaylibAMD_SceneTextureSize(SceneHandle scene, int32_t index, int32_t* outW, int32_t* outH);
RAYLIB_API void    RaylibAMD_SceneExportTexture(SceneHandle scene, int32_t index, float* outRGBA);
RAYLIB_API void    RaylibAMD_SceneGetSun(SceneHandle scene, float outIlluminance[3], float outDirection[3]);
/* BVH shape: nodes (64 B each), depth, and a host-side validity check (every triangle
 * inside its leaf's box, every child box inside its parent's).  Returns 1 if valid. */
RAYLIB_API int32_t RaylibAMD_SceneBVHInfo(SceneHandle scene, uint32_t* outNodes, uint32_t* outDepth, float* outSahCost);
/* The 4-wide collapse of the tree that the pool schedule traverses on large scenes: 0 = the scene has none (fewer than 8 triangles, or
 * analytic primitives), 1 = present and structurally valid (every triangle once, boxes nested, stack bound holds), -1 = invalid. */
RAYLIB_API int32_t RaylibAMD_SceneBVH4Info(SceneHandle scene, uint32_t* outNodes4, uint32_t* outWorstCaseStack);
/* The 8-wide collapse (80-byte grid nodes, children in octant order; scenes of more than 108 triangles): 0 = none, 1 = present and valid (every triangle slot
 * reached once, every node's 8-bit grid boxes contain the triangles below them, no path longer than *outLevels), -1 = invalid.  outSteps4 / outSteps8: the sum
 * over the 4-wide / 8-wide tree's nodes of (node area / root area) -- the node steps a random ray is expected to take; the megakernel walks the 8-wide
 * tree when outSteps4 >= 40 (RAYLIB_BVH8=0|1 overrides; RaylibAMDStats.treeWidth says which tree a frame walked). */
RAYLIB_API int32_t RaylibAMD_SceneBVH8Info(SceneHandle scene, uint32_t* outNodes8, uint32_t* outLevels, float* outSteps4, float* outSteps8);
/* The megakernel's walk of that tree restated on the host, operation by operation in float (csrc/rl_bvh.cc Walk8Host: the ray's per-node factors and error allowance, one fma per
 * 8-bit grid plane, octant visiting order, one stack entry per level), with the exit distance of ray i fixed at tMax[i]: outT[i] = the least distance among the triangles of the leaf
 * children the walk reaches (a tolerant double-precision triangle test; FLT_MAX: none), outSteps[i] (may be NULL) the node steps.  No device needed: what the box
 * arithmetic must never do -- skip the leaf of the closest hit -- is checked against the CPU oracle in `pytest -m "not gpu"`.  rays: count x (origin, direction).
 * Returns 1, 0 without such a tree, -1 on a malformed tree. */
RAYLIB_API int32_t RaylibAMD_SceneWalk8Host(SceneHandle scene, const float* rays, int32_t count, float tMin, const float* tMax, float* outT, uint32_t* outSteps);
/* The leaf list of a small scene (at most 24 leaves, 108 triangles): what k_trace walks instead of the tree when the scene is LDS-resident.
 * Returns the number of leaves (0 = the scene has none); the list's validity is part of RaylibAMD_SceneBVH4Info's check. */
RAYLIB_API int32_t RaylibAMD_SceneLeafListInfo(SceneHandle scene, uint32_t* outMaxTrianglesPerLeaf);
/* FNV-1a of the flat BVH (node records + leaf order): the multi-threaded build (RAYLIB_BUILD_THREADS, default = host
 * threads, <= 32) must give the tree of the single-threaded one. */
RAYLIB_API uint64_t RaylibAMD_SceneBVHHash(SceneHandle scene);
/* Camera derived state: 19 floats origin(3) lensRadius top_left(3) horizontal(3) vertical(3) u(3) v(3)
 * (reference render/camera.h:55-78). */
RAYLIB_API void    RaylibAMD_CameraExport(CameraHandle camera, float out[19]);
/* Create an image from caller memory (RGBA float, row 0 = top): lets tests use textures
 * and sky panoramas without an image codec. */
RAYLIB_API ImageHandle RaylibAMD_CreateImageFromData(uint32_t width, uint32_t height, const float* rgba);
/* Copy RGBA (4 floats per pixel) of an image to caller memory. */
RAYLIB_API void    RaylibAMD_DumpImageRGBA(ImageHandle image, float* outRGBA);
/* Test hook: the OBJ / MTL parser's number reader (value of strtof for one token). */
RAYLIB_API float   RaylibAMD_ParseFloat(const char* token);
/* Size of an image created by Raylib_LoadImage / Raylib_CreateImage (the reference ABI has no accessor). Returns 1 on success. */
RAYLIB_API int32_t RaylibAMD_ImageSize(ImageHandle image, uint32_t* outWidth, uint32_t* outHeight);
/* Replace a material's texture by an image handle (slot: 0 albedo, 1 normal, 2 roughness,
 * 3 metallic, 4 emissive) -- the OBJ loader does the same from map_* statements. */
RAYLIB_API int32_t RaylibAMD_OBJModelSetTexture(OBJModelHandle obj, const char* materialName, int32_t slot, ImageHandle image);

#ifdef __cplusplus
}
#endif
#endif /* RAYLIB_AMD_H */
