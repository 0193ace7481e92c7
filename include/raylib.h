/*
 * raylib.h -- the 33-function C-ABI of the raylib path tracer, served by the
 * MI355X-native library (software-raytracing_amd/libraylib.so).
 *
 * Each entry point replaces the reference function of the same name; the
 * citation is reference raylib/raylib.h:<decl> / raylib/raylib.cc:<impl>.
 * Behaviour kept: return codes (1/0, 0 handle on failure), ownership (the
 * library owns every object; a scene BORROWS OBJ models and the sky image),
 * synchronous Raylib_Render that resizes the output image to the viewport.
 * What changed behind it: Raylib_Render runs a HIP megakernel on gfx950
 * instead of a CPU thread pool (reference render/renderer.cc:273-356).
 */
#ifndef RAYLIB_H
#define RAYLIB_H

#include "raylib_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* raylib.h:24 / raylib.cc:25-41.  1 on success.  Fails (0) when no HIP device is usable. */
RAYLIB_API int32_t Raylib_Initialize(void);
/* raylib.h:27 / raylib.cc:43-51.  Returns 0, as the reference does. */
RAYLIB_API int32_t Raylib_Terminate(void);

/* raylib.h:35 / raylib.cc:56-69 */
RAYLIB_API OBJModelHandle Raylib_LoadOBJModel(const char* objPath);
/* raylib.h:37-41 / raylib.cc:71-90: rotate (yaw,pitch,roll degrees) -> scale -> translate */
RAYLIB_API void Raylib_TransformOBJModel(OBJModelHandle objModel,
	float translationX, float translationY, float translationZ,
	float yaw, float pitch, float roll,
	float scaleX, float scaleY, float scaleZ);
/* raylib.h:43 / raylib.cc:92-95 */
RAYLIB_API void Raylib_FinalizeOBJModel(OBJModelHandle objModel);
/* raylib.h:47 / raylib.cc:97-106 */
RAYLIB_API int32_t Raylib_UnloadOBJModel(OBJModelHandle objHandle);
/* raylib.h:52 / raylib.cc:108-113 */
RAYLIB_API ImageHandle Raylib_LoadImage(const char* filepath);

/* raylib.h:58 / raylib.cc:205-210 */
RAYLIB_API SceneHandle Raylib_CreateScene(void);
/* raylib.h:62 / raylib.cc:258-262 (element = C++ Hitable* in the reference) */
RAYLIB_API void Raylib_AddSceneElement(SceneHandle scene, SceneElementHandle element);
/* raylib.h:65 / raylib.cc:264-268 */
RAYLIB_API void Raylib_AddOBJModelToScene(SceneHandle scene, OBJModelHandle objModel);
/* raylib.h:67-69 / raylib.cc:270-283 */
RAYLIB_API void Raylib_SetSkyPanorama(SceneHandle scene, ImageHandle skyImage);
RAYLIB_API void Raylib_SetSunIlluminance(SceneHandle scene, float r, float g, float b);
RAYLIB_API void Raylib_SetSunDirection(SceneHandle scene, float x, float y, float z);
/* raylib.h:73 / raylib.cc:212-215 */
RAYLIB_API void Raylib_FinalizeScene(SceneHandle scene);
/* raylib.h:76 / raylib.cc:217-226 */
RAYLIB_API int32_t Raylib_DestroyScene(SceneHandle sceneHandle);

/* raylib.h:78-86 / raylib.cc:118-179 */
RAYLIB_API CameraHandle Raylib_CreateCamera(void);
RAYLIB_API void Raylib_CameraSetPosition(CameraHandle camera, float x, float y, float z);
RAYLIB_API void Raylib_CameraSetLookAt(CameraHandle camera, float tx, float ty, float tz);
RAYLIB_API void Raylib_CameraSetPerspective(CameraHandle camera, float fovY_degrees, float aspectWH);
RAYLIB_API void Raylib_CameraSetLens(CameraHandle camera, float aperture, float focalDistance);
RAYLIB_API void Raylib_CameraSetMotion(CameraHandle camera, float beginTime, float endTime);
RAYLIB_API void Raylib_CameraCopy(CameraHandle srcCamera, CameraHandle dstCamera);
RAYLIB_API int32_t Raylib_DestroyCamera(CameraHandle cameraHandle);

/* raylib.h:89 / raylib.cc:181-186 */
RAYLIB_API ImageHandle Raylib_CreateImage(uint32_t width, uint32_t height);
/* raylib.h:94 / raylib.cc:188-192: outDest holds 3*width*height floats, row-major RGB, row 0 = top */
RAYLIB_API void Raylib_DumpImageData(ImageHandle image, float* outDest);
/* raylib.h:97 / raylib.cc:194-203 */
RAYLIB_API int32_t Raylib_DestroyImage(ImageHandle imageHandle);

/* raylib.h:107-111 / raylib.cc:231-239 -- THE hot path. */
RAYLIB_API void Raylib_Render(const RendererSettings* settings, SceneHandle scene,
	CameraHandle camera, ImageHandle outMainImage);
/* raylib.h:121-126 / raylib.cc:241-256.  Returns 0 (no denoiser on this platform, as the reference on Linux). */
RAYLIB_API int32_t Raylib_Denoise(ImageHandle inMainImage, int32_t bMainImageHDR,
	ImageHandle inAlbedoImage, ImageHandle inNormalImage, ImageHandle outDenoisedImage);
/* raylib.h:130 / raylib.cc:285-288 */
RAYLIB_API void Raylib_PostProcess(ImageHandle image);
/* raylib.h:132 / raylib.cc:290-293 */
RAYLIB_API int32_t Raylib_IsDenoiserSupported(void);

/* raylib.h:139 / raylib.cc:298-312 */
RAYLIB_API const char* Raylib_GetRenderModeString(uint32_t auxMode);
/* raylib.h:147 / raylib.cc:314-326 */
RAYLIB_API int32_t Raylib_WriteImageToDisk(ImageHandle image, const char* filepath, uint32_t fileType);
/* raylib.h:150 / raylib.cc:328-331 */
RAYLIB_API void Raylib_FlushLogThread(void);

#ifdef __cplusplus
}
#endif
#endif /* RAYLIB_H */
