/*
 * TEST INFRASTRUCTURE -- flat scene description shared by the two checkers:
 *   oracle/ref_glue.cc  (drives the REAL reference sources -> oracle/_ref/*.so)
 *   oracle/oracle.cc    (CPU restatement of the reference algorithm)
 * and mirrored with ctypes in tests/.  The product library never includes this.
 *
 * It carries exactly what reference loader/obj_loader.cc:133-245 hands to the
 * renderer: triangles (positions, vertex normals, UVs, material, owning shape),
 * materials already mapped from MTL, float-RGBA textures, sun and sky.
 */
#ifndef ORACLE_FLAT_SCENE_H
#define ORACLE_FLAT_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct FlatTriangle {
	float v0[3], v1[3], v2[3];     /* reference geom/triangle.h:48-50 */
	float n0[3], n1[3], n2[3];     /* reference geom/triangle.h:55-57 */
	float s0, t0, s1, t1, s2, t2;  /* reference geom/triangle.h:62 (SetParameterization) */
	int32_t material;              /* index into FlatSceneDesc.materials */
	int32_t shape;                 /* StaticMesh index (one per OBJ shape, obj_loader.cc:133) */
} FlatTriangle;

enum FlatMaterialType {
	FLAT_MAT_LAMBERTIAN    = 0,    /* reference render/material.h:76-98  */
	FLAT_MAT_MIRROR        = 1,    /* reference render/material.h:142-169 */
	FLAT_MAT_DIELECTRIC    = 2,    /* reference render/material.h:123-140 */
	FLAT_MAT_MICROFACET    = 3,    /* reference render/material.h:171-270 */
	FLAT_MAT_METAL         = 4,    /* reference render/material.h:100-121 */
	FLAT_MAT_DIFFUSE_LIGHT = 5     /* reference render/material.h:50-74  */
};

typedef struct FlatMaterial {
	int32_t type;
	float albedo[3];        /* Lambertian albedo | Mirror baseColor | Microfacet albedoFallback | Metal albedo | DiffuseLight intensity */
	float roughness;        /* Microfacet roughnessFallback */
	float metallic;         /* Microfacet metallicFallback */
	float emissive[3];      /* Microfacet emissiveFallback */
	float ior;              /* Dielectric ref_idx */
	float transmission[3];  /* Dielectric transmissionFilter */
	float fuzziness;        /* Metal */
	int32_t texAlbedo;      /* texture indices, -1 = none */
	int32_t texNormal;
	int32_t texRoughness;
	int32_t texMetallic;
	int32_t texEmissive;
} FlatMaterial;

typedef struct FlatTexture {
	int32_t width, height;
	const float* rgba;      /* row-major, row 0 = top, 4 floats per texel (reference render/image.h:88-119) */
} FlatTexture;

typedef struct FlatCamera {
	float origin[3];
	float lookAt[3];
	float fovY_degrees, aspectWH;
	float aperture, focalDistance;
	float beginTime, endTime;
} FlatCamera;

/* Layout-identical to reference raylib_types.h:41-57 (24 bytes). */
typedef struct FlatSettings {
	uint32_t viewportWidth, viewportHeight;
	int32_t  samplesPerPixel, maxPathLength;
	float    rayTMin;
	uint32_t renderMode;
} FlatSettings;

/* Analytic primitives added with Raylib_AddSceneElement in the reference's procedural scenes
 * (reference src/main.cc:913-984): geom/sphere.h:8-26, geom/cube.h:8-42. */
typedef struct FlatSphere { float center[3]; float radius; int32_t material; } FlatSphere;
typedef struct FlatCube { float minBounds[3]; float maxBounds[3]; float timeStartMove; float velocity[3]; int32_t material; } FlatCube;

typedef struct FlatSceneDesc {
	const FlatTriangle* triangles; int32_t numTriangles;
	const FlatMaterial* materials; int32_t numMaterials;
	const FlatTexture*  textures;  int32_t numTextures;
	int32_t numShapes;
	float sunIlluminance[3];
	float sunDirection[3];      /* un-normalised; Scene normalises (reference geom/scene.h:20) */
	int32_t skyTexture;         /* -1 = none */
	/* scene elements are added in this order: the OBJ root (if numTriangles > 0), the spheres, the cubes */
	const FlatSphere* spheres; int32_t numSpheres;
	const FlatCube* cubes;     int32_t numCubes;
} FlatSceneDesc;

/* Closest-hit record returned by both checkers (reference geom/hit.h:16-36). */
typedef struct FlatHit {
	int32_t hit;
	float t;
	float p[3];
	float n[3];
	float paramU, paramV;
	int32_t material;
} FlatHit;

#ifdef __cplusplus
}
#endif
#endif
