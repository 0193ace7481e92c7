"""ctypes binding of libraylib.so -- the Python mirror of the reference's own FFI
wrapper (gui-app/gui-app/RaylibWrapper.cs:43-145 binds the same 33 functions with
P/Invoke).  Function names, argument order and return conventions are the C-ABI's.

There is no fallback: if the shared library is missing this module raises, and if
no HIP device is present Raylib_Initialize returns 0 and Raylib_Render fails loudly.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAYLIB_LIB") or os.path.join(os.path.dirname(HERE), "libraylib.so")

RENDERMODE_DEFAULT, RENDERMODE_ALBEDO, RENDERMODE_SURFACE_NORMAL, RENDERMODE_MICROSURFACE_NORMAL, \
    RENDERMODE_TEXCOORD, RENDERMODE_EMISSION, RENDERMODE_REFLECTANCE = range(7)


class RendererSettings(C.Structure):
    """reference raylib_types.h:41-57 / RaylibWrapper.cs:27-38 (24 bytes)."""
    _fields_ = [("viewportWidth", C.c_uint32), ("viewportHeight", C.c_uint32),
                ("samplesPerPixel", C.c_int32), ("maxPathLength", C.c_int32),
                ("rayTMin", C.c_float), ("renderMode", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodesVisited", C.c_uint64), ("trisTested", C.c_uint64),
                ("shadedHits", C.c_uint64), ("texFetches", C.c_uint64), ("cameraSamples", C.c_uint64),
                ("pixels", C.c_uint64), ("kernelMs", C.c_double), ("traceKernelMs", C.c_double), ("wallMs", C.c_double),
                ("traceLaunches", C.c_uint32), ("numNodes", C.c_uint32), ("numTriangles", C.c_uint32), ("bvhDepth", C.c_uint32),
                ("waveTrips", C.c_uint64), ("pathsPerWave", C.c_uint32), ("ranks", C.c_uint32),
                ("gatherMode", C.c_uint32), ("rcclCommSize", C.c_uint32), ("devices", C.c_uint32), ("jobHeads", C.c_uint32),
                ("gatherMs", C.c_double), ("scatterMs", C.c_double), ("rankKernelMs", C.c_double * 16), ("rankTraceMs", C.c_double * 16),
                ("culledCells", C.c_uint32), ("listedCells", C.c_uint32), ("culledSamples", C.c_uint64), ("culledRays", C.c_uint64),
                ("treeWidth", C.c_uint32), ("nodeBytes", C.c_uint32)]

    GATHER_MODES = {0: "none", 1: "rccl", 2: "peer"}

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["rankKernelMs"] = list(self.rankKernelMs)[: max(1, self.ranks)]
        d["rankTraceMs"] = list(self.rankTraceMs)[: max(1, self.ranks)]
        d["gatherMode"] = self.GATHER_MODES.get(self.gatherMode, "?")
        # the frame's totals, whatever share of it the kernels had to execute: equal with and without the silhouette cull (csrc/rl_cull.cc)
        d["frameSamples"] = self.cameraSamples + self.culledSamples
        d["frameRays"] = self.rays + self.culledRays
        d["frameNodes"] = self.nodesVisited + self.culledRays
        return d


# Per-ray / per-unit algorithmic byte constants of the flat layout (csrc/rl_device.h)
NODE_B, TRI_B, SHADE_B, TEXEL_B, PIXEL_B = 64, 64, 64, 16, 16


def algorithmic_bytes(stats):
    """nodes*NODE_B + tris*TRI_B + shaded*SHADE_B + texels*TEXEL_B + pixels*16 (NODE_B = 80 when the 8-wide tree was walked)."""
    return (stats.nodesVisited * (getattr(stats, "nodeBytes", 0) or NODE_B) + stats.trisTested * TRI_B + stats.shadedHits * SHADE_B +
            stats.texFetches * TEXEL_B + stats.pixels * PIXEL_B)


_EXPORTS = {
    # name: (restype, argtypes)   -- include/raylib.h
    "Raylib_Initialize": (C.c_int32, []),
    "Raylib_Terminate": (C.c_int32, []),
    "Raylib_LoadOBJModel": (C.c_void_p, [C.c_char_p]),
    "Raylib_TransformOBJModel": (None, [C.c_void_p] + [C.c_float] * 9),
    "Raylib_FinalizeOBJModel": (None, [C.c_void_p]),
    "Raylib_UnloadOBJModel": (C.c_int32, [C.c_void_p]),
    "Raylib_LoadImage": (C.c_void_p, [C.c_char_p]),
    "Raylib_CreateScene": (C.c_void_p, []),
    "Raylib_AddSceneElement": (None, [C.c_void_p, C.c_void_p]),
    "Raylib_AddOBJModelToScene": (None, [C.c_void_p, C.c_void_p]),
    "Raylib_SetSkyPanorama": (None, [C.c_void_p, C.c_void_p]),
    "Raylib_SetSunIlluminance": (None, [C.c_void_p, C.c_float, C.c_float, C.c_float]),
    "Raylib_SetSunDirection": (None, [C.c_void_p, C.c_float, C.c_float, C.c_float]),
    "Raylib_FinalizeScene": (None, [C.c_void_p]),
    "Raylib_DestroyScene": (C.c_int32, [C.c_void_p]),
    "Raylib_CreateCamera": (C.c_void_p, []),
    "Raylib_CameraSetPosition": (None, [C.c_void_p, C.c_float, C.c_float, C.c_float]),
    "Raylib_CameraSetLookAt": (None, [C.c_void_p, C.c_float, C.c_float, C.c_float]),
    "Raylib_CameraSetPerspective": (None, [C.c_void_p, C.c_float, C.c_float]),
    "Raylib_CameraSetLens": (None, [C.c_void_p, C.c_float, C.c_float]),
    "Raylib_CameraSetMotion": (None, [C.c_void_p, C.c_float, C.c_float]),
    "Raylib_CameraCopy": (None, [C.c_void_p, C.c_void_p]),
    "Raylib_DestroyCamera": (C.c_int32, [C.c_void_p]),
    "Raylib_CreateImage": (C.c_void_p, [C.c_uint32, C.c_uint32]),
    "Raylib_DumpImageData": (None, [C.c_void_p, C.POINTER(C.c_float)]),
    "Raylib_DestroyImage": (C.c_int32, [C.c_void_p]),
    "Raylib_Render": (None, [C.POINTER(RendererSettings), C.c_void_p, C.c_void_p, C.c_void_p]),
    "Raylib_Denoise": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "Raylib_PostProcess": (None, [C.c_void_p]),
    "Raylib_IsDenoiserSupported": (C.c_int32, []),
    "Raylib_GetRenderModeString": (C.c_char_p, [C.c_uint32]),
    "Raylib_WriteImageToDisk": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_uint32]),
    "Raylib_FlushLogThread": (None, []),
    # include/raylib_amd.h
    "RaylibAMD_SetSeed": (None, [C.c_uint64]),
    "RaylibAMD_GetSeed": (C.c_uint64, []),
    "RaylibAMD_GetLastStats": (None, [C.POINTER(Stats)]),
    "RaylibAMD_DeviceAvailable": (C.c_int32, []),
    "RaylibAMD_BuildId": (C.c_char_p, []),
    "RaylibAMD_RenderDevice": (C.c_int32, [C.POINTER(RendererSettings), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "RaylibAMD_RenderCellsHost": (C.c_int32, [C.POINTER(RendererSettings), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "RaylibAMD_CellBufferFloats": (C.c_uint64, [C.c_uint32] * 4),
    "RaylibAMD_NumCells": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "RaylibAMD_CreateMaterial": (C.c_void_p, [C.c_int32, C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float), C.c_float]),
    "RaylibAMD_DestroyMaterial": (C.c_int32, [C.c_void_p]),
    "RaylibAMD_CreateSphere": (C.c_void_p, [C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "RaylibAMD_CreateCube": (C.c_void_p, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float), C.c_void_p]),
    "RaylibAMD_CreateTriangle": (C.c_void_p, [C.POINTER(C.c_float)] * 7 + [C.c_void_p]),
    "RaylibAMD_DestroySceneElement": (C.c_int32, [C.c_void_p]),
    "RaylibAMD_EvalScatter": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.c_int32, C.c_uint64, C.POINTER(C.c_float)]),
    "RaylibAMD_EvalCameraRays": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_uint64, C.POINTER(C.c_float)]),
    "RaylibAMD_EvalTexture": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float)]),
    "RaylibAMD_EvalDeviceMath": (C.c_int32, [C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float)]),
    "RaylibAMD_ClosestHit": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_float, C.c_void_p]),
    "RaylibAMD_VerifyExactMath": (C.c_int32, [C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "RaylibAMD_CullCells": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.POINTER(C.c_float)]),
    "RaylibAMD_SceneNumTriangles": (C.c_int32, [C.c_void_p]),
    "RaylibAMD_SceneNumMaterials": (C.c_int32, [C.c_void_p]),
    "RaylibAMD_SceneNumTextures": (C.c_int32, [C.c_void_p]),
    "RaylibAMD_SceneExportTriangles": (None, [C.c_void_p, C.c_void_p]),
    "RaylibAMD_SceneExportMaterials": (None, [C.c_void_p, C.c_void_p]),
    "RaylibAMD_SceneTextureSize": (None, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "RaylibAMD_SceneExportTexture": (None, [C.c_void_p, C.c_int32, C.POINTER(C.c_float)]),
    "RaylibAMD_SceneGetSun": (None, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "RaylibAMD_SceneBVHInfo": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "RaylibAMD_ParseFloat": (C.c_float, [C.c_char_p]),
    "RaylibAMD_ImageSize": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "RaylibAMD_SceneBVH4Info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "RaylibAMD_SceneBVH8Info": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "RaylibAMD_SceneLeafListInfo": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "RaylibAMD_SceneWalk8Host": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "RaylibAMD_SceneBVHHash": (C.c_uint64, [C.c_void_p]),
    "RaylibAMD_CameraExport": (None, [C.c_void_p, C.POINTER(C.c_float)]),
    "RaylibAMD_CreateImageFromData": (C.c_void_p, [C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "RaylibAMD_DumpImageRGBA": (None, [C.c_void_p, C.POINTER(C.c_float)]),
    "RaylibAMD_OBJModelSetTexture": (C.c_int32, [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p]),
}
RAYLIB_H_EXPORTS = [k for k in _EXPORTS if k.startswith("Raylib_")]
RAYLIB_AMD_H_EXPORTS = [k for k in _EXPORTS if k.startswith("RaylibAMD_")]


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise FileNotFoundError(path + " -- build it with `make -C software-raytracing_amd` (or __graft_entry__.build())")
    lib = C.CDLL(path)
    for name, (res, args) in _EXPORTS.items():
        if os.environ.get("RAYLIB_LIB") and name.startswith("RaylibAMD_") and not hasattr(lib, name):
            continue              # (an older build named by RAYLIB_LIB for an A/B run may lack a newer introspection hook; the tree's own library must export everything)
        fn = getattr(lib, name)   # AttributeError if the library does not export it
        fn.restype, fn.argtypes = res, args
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def create_material(lib, mat):
    """mat: one record of the oracle's MAT_DTYPE layout (type, albedo, roughness, metallic, emissive, ior, transmission, fuzziness)."""
    return lib.RaylibAMD_CreateMaterial(int(mat["type"]), _f3(mat["albedo"]), float(mat["roughness"]), float(mat["metallic"]),
                                        _f3(mat["emissive"]), float(mat["ior"]), _f3(mat["transmission"]), float(mat["fuzziness"]))


class ProceduralSession:
    """A scene made of analytic elements through the C-ABI (what the reference's CUI does with C++ objects,
    src/main.cc:913-984): materials -> spheres / cubes -> Raylib_AddSceneElement -> FinalizeScene."""

    def __init__(self, lib, materials, spheres=(), cubes=(), origin=(0, 0, 3), look_at=(0, 0, -1), fov=45.0, aspect=1.0,
                 sun=(0, 0, 0), sun_dir=(0.0, -1.0, -0.5), aperture=0.0, focal=1.0, shutter=(0.0, 0.0)):
        self.lib = lib
        self.mats = [create_material(lib, m) for m in materials]
        assert all(self.mats)
        self.scene = lib.Raylib_CreateScene()
        self.camera = lib.Raylib_CreateCamera()
        self.elems = []
        for s in spheres:
            e = lib.RaylibAMD_CreateSphere(float(s["center"][0]), float(s["center"][1]), float(s["center"][2]), float(s["radius"]), self.mats[int(s["material"])])
            self.elems.append(e); lib.Raylib_AddSceneElement(self.scene, e)
        for c in cubes:
            e = lib.RaylibAMD_CreateCube(_f3(c["minBounds"]), _f3(c["maxBounds"]), float(c["timeStartMove"]), _f3(c["velocity"]), self.mats[int(c["material"])])
            self.elems.append(e); lib.Raylib_AddSceneElement(self.scene, e)
        assert all(self.elems)
        lib.Raylib_SetSunIlluminance(self.scene, *[float(x) for x in sun])
        lib.Raylib_SetSunDirection(self.scene, *[float(x) for x in sun_dir])
        lib.Raylib_FinalizeScene(self.scene)
        lib.Raylib_CameraSetPosition(self.camera, *[float(x) for x in origin])
        lib.Raylib_CameraSetLookAt(self.camera, *[float(x) for x in look_at])
        lib.Raylib_CameraSetPerspective(self.camera, float(fov), float(aspect))
        lib.Raylib_CameraSetLens(self.camera, float(aperture), float(focal))
        lib.Raylib_CameraSetMotion(self.camera, float(shutter[0]), float(shutter[1]))

    settings = None

    def render(self, w, h, spp, max_path=5, tmin=1e-4, mode=RENDERMODE_DEFAULT):
        lib = self.lib
        st = RendererSettings(int(w), int(h), int(spp), int(max_path), float(tmin), int(mode))
        img = lib.Raylib_CreateImage(w, h)
        lib.Raylib_Render(C.byref(st), self.scene, self.camera, img)
        out = np.zeros((h, w, 4), np.float32)
        lib.RaylibAMD_DumpImageRGBA(img, _fp(out))
        lib.Raylib_DestroyImage(img)
        return out

    def close(self):
        lib = self.lib
        lib.Raylib_DestroyScene(self.scene); lib.Raylib_DestroyCamera(self.camera)
        for e in self.elems:
            lib.RaylibAMD_DestroySceneElement(e)
        for m in self.mats:
            lib.RaylibAMD_DestroyMaterial(m)


class SceneSession:
    """The GUI's call sequence (reference gui-app/gui-app/MainForm.cs:121-256) as an object:
    LoadOBJ -> FinalizeOBJ -> CreateScene/Camera -> AddOBJ -> Sun -> FinalizeScene -> camera setters."""

    def __init__(self, lib, obj_path, origin, look_at, fov, aspect, sun=(0, 0, 0), sun_dir=(0.0, -1.0, -0.5),
                 aperture=0.0, focal=1.0, shutter=(0.0, 0.0), sky_image=None, textures=()):
        self.lib = lib
        self.obj = lib.Raylib_LoadOBJModel(obj_path.encode())
        if not self.obj:
            raise RuntimeError("Raylib_LoadOBJModel failed: " + obj_path)
        self._images = []
        for mat_name, slot, rgba in textures:
            rgba = np.ascontiguousarray(rgba, np.float32)
            ih = lib.RaylibAMD_CreateImageFromData(rgba.shape[1], rgba.shape[0], _fp(rgba))
            self._images.append(ih)
            if not lib.RaylibAMD_OBJModelSetTexture(self.obj, mat_name.encode(), slot, ih):
                raise RuntimeError("RaylibAMD_OBJModelSetTexture failed for " + mat_name)
        lib.Raylib_FinalizeOBJModel(self.obj)
        self.scene = lib.Raylib_CreateScene()
        self.camera = lib.Raylib_CreateCamera()
        lib.Raylib_AddOBJModelToScene(self.scene, self.obj)
        lib.Raylib_SetSunIlluminance(self.scene, *[float(x) for x in sun])
        lib.Raylib_SetSunDirection(self.scene, *[float(x) for x in sun_dir])
        if sky_image is not None:
            sky = np.ascontiguousarray(sky_image, np.float32)
            ih = lib.RaylibAMD_CreateImageFromData(sky.shape[1], sky.shape[0], _fp(sky))
            self._images.append(ih)
            lib.Raylib_SetSkyPanorama(self.scene, ih)
        lib.Raylib_FinalizeScene(self.scene)
        lib.Raylib_CameraSetPosition(self.camera, *[float(x) for x in origin])
        lib.Raylib_CameraSetLookAt(self.camera, *[float(x) for x in look_at])
        lib.Raylib_CameraSetPerspective(self.camera, float(fov), float(aspect))
        lib.Raylib_CameraSetLens(self.camera, float(aperture), float(focal))
        lib.Raylib_CameraSetMotion(self.camera, float(shutter[0]), float(shutter[1]))

    def settings(self, w, h, spp, max_path=5, tmin=1e-4, mode=RENDERMODE_DEFAULT):
        return RendererSettings(int(w), int(h), int(spp), int(max_path), float(tmin), int(mode))

    def render(self, w, h, spp, max_path=5, tmin=1e-4, mode=RENDERMODE_DEFAULT):
        """Raylib_Render into a fresh image; returns (H, W, 4) float32 RGBA."""
        lib = self.lib
        st = self.settings(w, h, spp, max_path, tmin, mode)
        img = lib.Raylib_CreateImage(w, h)
        lib.Raylib_Render(C.byref(st), self.scene, self.camera, img)
        out = np.zeros((h, w, 4), np.float32)
        lib.RaylibAMD_DumpImageRGBA(img, _fp(out))
        lib.Raylib_DestroyImage(img)
        return out

    def render_cells(self, w, h, spp, rank, world, max_path=5, tmin=1e-4, mode=RENDERMODE_DEFAULT):
        """What rank `rank` of `world` renders: its cells back to back, (n_cells*64, 4) float32."""
        st = self.settings(w, h, spp, max_path, tmin, mode)
        n = self.lib.RaylibAMD_CellBufferFloats(w, h, rank, world)
        out = np.zeros(max(n, 4), np.float32)
        if self.lib.RaylibAMD_RenderCellsHost(C.byref(st), self.scene, self.camera, rank, world, _fp(out)) != 1:
            raise RuntimeError("RaylibAMD_RenderCellsHost failed")
        return out[:n].reshape(-1, 4)

    def stats(self):
        s = Stats()
        self.lib.RaylibAMD_GetLastStats(C.byref(s))
        return s

    def export_flat(self):
        """(triangles, materials) as numpy arrays with the oracle's record layouts."""
        from_tri = np.dtype([("v0", "f4", 3), ("v1", "f4", 3), ("v2", "f4", 3), ("n0", "f4", 3), ("n1", "f4", 3), ("n2", "f4", 3),
                             ("st", "f4", 6), ("material", "i4"), ("shape", "i4")])
        from_mat = np.dtype([("type", "i4"), ("albedo", "f4", 3), ("roughness", "f4"), ("metallic", "f4"), ("emissive", "f4", 3),
                             ("ior", "f4"), ("transmission", "f4", 3), ("fuzziness", "f4"),
                             ("texAlbedo", "i4"), ("texNormal", "i4"), ("texRoughness", "i4"), ("texMetallic", "i4"), ("texEmissive", "i4")])
        lib = self.lib
        tris = np.zeros(lib.RaylibAMD_SceneNumTriangles(self.scene), from_tri)
        mats = np.zeros(lib.RaylibAMD_SceneNumMaterials(self.scene), from_mat)
        lib.RaylibAMD_SceneExportTriangles(self.scene, tris.ctypes.data)
        lib.RaylibAMD_SceneExportMaterials(self.scene, mats.ctypes.data)
        return tris, mats

    def close(self):
        lib = self.lib
        # reference order: model, scene, camera, images (MainForm.cs:253-256)
        lib.Raylib_UnloadOBJModel(self.obj)
        lib.Raylib_DestroyScene(self.scene)
        lib.Raylib_DestroyCamera(self.camera)
        for ih in self._images:
            lib.Raylib_DestroyImage(ih)
