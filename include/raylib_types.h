/*
 * raylib_types.h -- handle types, enums and RendererSettings of the raylib C-ABI.
 * Drop-in for reference raylib/raylib_types.h:13-57 (same names, values, layout;
 * the C# front-end mirrors them in gui-app/gui-app/RaylibWrapper.cs:5-38).
 */
#ifndef RAYLIB_TYPES_H
#define RAYLIB_TYPES_H

#include <stdint.h>

#if defined(_WIN32)
  #ifdef RAYLIB_EXPORTS
    #define RAYLIB_API __declspec(dllexport)
  #else
    #define RAYLIB_API __declspec(dllimport)
  #endif
#else
  #define RAYLIB_API __attribute__((visibility("default")))
#endif

/* Opaque handles; 0 means failure (reference raylib_types.h:13-17). */
typedef uintptr_t OBJModelHandle;
typedef uintptr_t ImageHandle;
typedef uintptr_t SceneHandle;
typedef uintptr_t SceneElementHandle;
typedef uintptr_t CameraHandle;

/* reference raylib_types.h:19-30 */
enum ERenderMode {
	RAYLIB_RENDERMODE_Default            = 0,
	RAYLIB_RENDERMODE_Albedo             = 1,
	RAYLIB_RENDERMODE_SurfaceNormal      = 2,
	RAYLIB_RENDERMODE_MicrosurfaceNormal = 3,
	RAYLIB_RENDERMODE_Texcoord           = 4,
	RAYLIB_RENDERMODE_Emission           = 5,
	RAYLIB_RENDERMODE_Reflectance        = 6,
	RAYLIB_RENDERMODE_MAX
};

/* reference raylib_types.h:32-39 */
enum EImageFileType {
	RAYLIB_IMAGEFILETYPE_Bitmap = 0,
	RAYLIB_IMAGEFILETYPE_Jpg    = 1,
	RAYLIB_IMAGEFILETYPE_Png    = 2,
	RAYLIB_IMAGEFILETYPE_MAX
};

/* reference raylib_types.h:41-57 -- 24 bytes, sequential; never extended
 * (seed / GPU selection travel through env vars or raylib_amd.h instead). */
typedef struct RendererSettings {
	uint32_t viewportWidth;
	uint32_t viewportHeight;
	int32_t  samplesPerPixel;
	int32_t  maxPathLength;
	float    rayTMin;
	uint32_t renderMode;
#ifdef __cplusplus
	inline float getViewportAspectWH() const { return (float)viewportWidth / (float)viewportHeight; }
#endif
} RendererSettings;

#endif /* RAYLIB_TYPES_H */
