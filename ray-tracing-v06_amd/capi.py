"""ctypes binding of librt06.so (the C ABI declared in include/rt06.h).

The product path is the HIP library and nothing else: if librt06.so is missing or fails to load this
raises — there is no CPU fallback and nothing here imports the oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
# RT06_LIB: development only — another build of the SAME library (e.g. one compiled with -DRT_PHASE_TIMERS); never a fallback
LIB_PATH = os.environ.get("RT06_LIB") or os.path.join(CSRC_DIR, "librt06.so")

RT_OK = 0
RT_PRIM_MOVING = 0x80000000
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_LAMBERTIAN_CHECKER, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC, MAT_LAMBERTIAN_NOISE, MAT_LAMBERTIAN_IMAGE = 0, 1, 2, 3, 4, 5, 6, 7
WORLD_BVH, WORLD_LIST, WORLD_NODE_TREE = 0, 1, 2
CAM_PINHOLE, CAM_DEFOCUS, CAM_MOTION = 0, 1, 2

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
vec3 = C.c_float * 3

NODE_DT = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("left", "<i4"), ("right", "<i4")])
PRIM_DT = np.dtype([("c0", "<f4", 3), ("radius", "<f4"), ("c1", "<f4", 3), ("mat", "<u4")])
MAT_DT = np.dtype([("albedo", "<f4", 3), ("param", "<f4"), ("albedo2", "<f4", 3), ("type", "<u4")])
QUAD_DT = np.dtype([("Q", "<f4", 3), ("D", "<f4"), ("u", "<f4", 3), ("mat", "<u4"), ("v", "<f4", 3), ("pad0", "<f4"),
                    ("normal", "<f4", 3), ("pad1", "<f4"), ("w", "<f4", 3), ("pad2", "<f4")])


class WorldFlat(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("root", C.c_int32), ("n_nodes", C.c_uint32), ("n_prims", C.c_uint32),
                ("n_materials", C.c_uint32), ("max_stack", C.c_uint32),
                ("bounds_min", vec3), ("bounds_max", vec3),
                ("nodes", C.c_void_p), ("prims", C.c_void_p), ("materials", C.c_void_p),
                ("quads", C.c_void_p), ("n_quads", C.c_uint32), ("background", C.c_uint32),
                ("background_color", vec3), ("image_width", C.c_uint32),
                ("perlin", C.c_void_p), ("image", C.c_void_p), ("image_height", C.c_uint32), ("traversal", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("type", C.c_uint32), ("o", vec3), ("u", vec3), ("v", vec3), ("w", vec3),
                ("viewport_width", C.c_float), ("viewport_height", C.c_float),
                ("lens_radius", C.c_float), ("focus_dist", C.c_float), ("t0", C.c_float), ("t1", C.c_float)]


class RenderConfig(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_pixel", C.c_uint32),
                ("max_depth", C.c_uint32), ("seed", C.c_uint64), ("device", C.c_int32),
                ("rank", C.c_uint32), ("world_size", C.c_uint32), ("variant", C.c_uint32)]


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt06 error {code}: {msg}")
        self.code = code


# every symbol include/rt06.h declares (tests check the library exports all of them)
SYMBOLS = [
    "rt_last_error", "rt_camera_pinhole", "rt_camera_defocus", "rt_camera_motion",
    "rt_scene_create", "rt_scene_destroy", "rt_scene_add_material", "rt_scene_add_sphere",
    "rt_scene_add_moving_sphere", "rt_scene_add_quad", "rt_scene_set_background", "rt_scene_set_perlin", "rt_scene_set_image", "rt_scene_cornell_box", "rt_scene_add_box", "rt_scene_book2_final", "rt_scene_prim_bounds", "rt_scene_build_bvh_topdown", "rt_scene_build_bvh_sah",
    "rt_scene_build_bvh_bottomup", "rt_scene_set_world_list", "rt_scene_add_bvh_node",
    "rt_scene_set_world_node_tree", "rt_scene_get_flat", "rt_scene_book1_final", "rt_scene_book2_moving",
    "rt_scene_three_spheres", "rt_host_uniforms", "rt_renderer_create", "rt_renderer_destroy", "rt_renderer_render",
    "rt_renderer_render_async", "rt_renderer_last_kernel_ms", "rt_renderer_kernel_info", "rt_renderer_download", "rt_renderer_shard_floats",
    "rt_renderer_assemble", "rt_renderer_kernel_times", "rt_multi_renderer_create", "rt_multi_renderer_destroy", "rt_multi_renderer_render",
    "rt_multi_renderer_download", "rt_multi_renderer_times", "rt_multi_renderer_gpus", "rt_shard_layout", "rt_shard_pixel_map", "rt_device_info", "rt_scene_set_traversal", "rt_probe_aabb", "rt_probe_sphere", "rt_probe_trace", "rt_probe_scatter",
    "rt_probe_camera", "rt_probe_radiance", "rt_probe_sphere_index", "rt_probe_rng", "rt_probe_math", "rt_probe_glm", "rt_probe_aabb_misc", "rt_probe_aabb_regular", "rt_probe_boxpair_filtered", "rt_probe_boxpair_certified",
    "rt_selftest_fastdiv", "rt_selftest_fastdiv4", "rt_selftest_fastrcp", "rt_device_count", "rt_version", "rt_source_hash", "rt_renderer_pass_info",
]

_lib = None


def source_hash():
    """sha256 over the sources librt06.so is built from, exactly as csrc/Makefile computes it for the stamp it embeds in the
    library (SRCS, HDRS, the Makefile, the EXTRA flags of a plain build): the hash a fresh `make` of this tree WOULD embed."""
    import hashlib
    import re
    mk = open(os.path.join(CSRC_DIR, "Makefile")).read()
    names = re.search(r"^SRCS\s*:=\s*(.*)$", mk, re.M).group(1).split() + re.search(r"^HDRS\s*:=\s*(.*)$", mk, re.M).group(1).split()
    files = [os.path.join(os.path.dirname(PKG_DIR), "include", "rt06.h") if n.endswith("rt06.h") else os.path.join(CSRC_DIR, n) for n in names]
    files.append(os.path.join(CSRC_DIR, "Makefile"))
    h = hashlib.sha256()
    for f in files:
        h.update(open(f, "rb").read())
    h.update(b"\n")   # echo "$(EXTRA)" of a plain build
    return h.hexdigest()


def library_hash():
    """The source hash embedded in the librt06.so that is actually loaded (rt_source_hash): what the running BINARY was built from."""
    return lib().rt_source_hash().decode()


def build_native(force=False):
    """Compile librt06.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", CSRC_DIR]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return LIB_PATH


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so with the same SONAME as /opt/rocm's.  A process
    must run on ONE HIP runtime, and torch only works on its own, so when torch is installed its copy is
    loaded first (without importing torch); librt06.so's libamdhip64.so.7 dependency then binds to it and
    device pointers / streams can be shared with torch whichever of the two is imported first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(hipcc --offload-arch=gfx950). There is no CPU fallback for the render path.")
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    L.rt_last_error.restype = C.c_char_p
    L.rt_version.restype = C.c_char_p
    L.rt_source_hash.restype = C.c_char_p
    L.rt_renderer_pass_info.argtypes = [C.c_void_p, C.c_uint64 * 4]
    L.rt_camera_pinhole.argtypes = [vec3, vec3, vec3, C.c_float, C.c_float, P(Camera)]
    L.rt_camera_defocus.argtypes = [vec3, vec3, vec3, C.c_float, C.c_float, C.c_float, C.c_float, P(Camera)]
    L.rt_camera_motion.argtypes = [vec3, vec3, vec3, C.c_float, C.c_float, C.c_float, C.c_float, P(Camera)]
    L.rt_scene_create.argtypes = [P(C.c_void_p)]
    L.rt_scene_destroy.argtypes = [C.c_void_p]
    L.rt_scene_destroy.restype = None
    L.rt_scene_add_material.argtypes = [C.c_void_p, C.c_uint32, vec3, C.c_float, C.c_void_p, P(C.c_int32)]
    L.rt_scene_add_sphere.argtypes = [C.c_void_p, vec3, C.c_float, C.c_int32, P(C.c_int32)]
    L.rt_scene_add_moving_sphere.argtypes = [C.c_void_p, vec3, vec3, C.c_float, C.c_int32, P(C.c_int32)]
    L.rt_scene_add_quad.argtypes = [C.c_void_p, vec3, vec3, vec3, C.c_int32, P(C.c_int32)]
    L.rt_scene_set_background.argtypes = [C.c_void_p, C.c_uint32, vec3]
    L.rt_scene_set_perlin.argtypes = [C.c_void_p, C.c_uint64]
    L.rt_scene_set_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.rt_scene_cornell_box.argtypes = [P(C.c_void_p)]
    L.rt_scene_add_box.argtypes = [C.c_void_p, vec3, vec3, C.c_int32, C.c_float, vec3, P(C.c_int32)]
    L.rt_scene_book2_final.argtypes = [C.c_uint64, P(C.c_void_p)]
    L.rt_scene_prim_bounds.argtypes = [C.c_void_p, C.c_int32, vec3, vec3]
    for n in ("rt_scene_build_bvh_topdown", "rt_scene_build_bvh_sah", "rt_scene_build_bvh_bottomup", "rt_scene_set_world_list"):
        getattr(L, n).argtypes = [C.c_void_p]
    L.rt_scene_add_bvh_node.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, P(C.c_int32)]
    L.rt_scene_set_world_node_tree.argtypes = [C.c_void_p, C.c_int32]
    L.rt_scene_get_flat.argtypes = [C.c_void_p, P(WorldFlat)]
    L.rt_scene_book1_final.argtypes = [C.c_uint64, P(C.c_void_p)]
    L.rt_scene_book2_moving.argtypes = [C.c_uint64, P(C.c_void_p)]
    L.rt_scene_three_spheres.argtypes = [P(C.c_void_p)]
    L.rt_host_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, f32p]
    L.rt_renderer_create.argtypes = [P(RenderConfig), P(Camera), P(WorldFlat), P(C.c_void_p)]
    L.rt_renderer_destroy.argtypes = [C.c_void_p]
    L.rt_renderer_destroy.restype = None
    L.rt_renderer_render.argtypes = [C.c_void_p]
    L.rt_renderer_render_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_renderer_last_kernel_ms.argtypes = [C.c_void_p, P(C.c_float)]
    L.rt_renderer_kernel_info.argtypes = [C.c_void_p, P(C.c_uint32 * 4)]
    L.rt_renderer_kernel_times.argtypes = [C.c_void_p, C.c_uint32, C.c_float * 3]
    L.rt_multi_renderer_create.argtypes = [P(RenderConfig), P(Camera), P(WorldFlat), C.c_uint32, C.c_void_p, P(C.c_void_p)]
    L.rt_multi_renderer_destroy.argtypes = [C.c_void_p]
    L.rt_multi_renderer_destroy.restype = None
    L.rt_multi_renderer_render.argtypes = [C.c_void_p]
    L.rt_multi_renderer_download.argtypes = [C.c_void_p, f32p, C.c_size_t]
    L.rt_multi_renderer_times.argtypes = [C.c_void_p, C.c_float * 3]
    L.rt_multi_renderer_gpus.argtypes = [C.c_void_p, P(C.c_uint32)]
    L.rt_device_info.argtypes = [C.c_int32, C.c_uint32 * 4]
    L.rt_scene_set_traversal.argtypes = [C.c_void_p, C.c_uint32]
    L.rt_shard_layout.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32 * 4]
    L.rt_shard_pixel_map.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_size_t]
    L.rt_renderer_download.argtypes = [C.c_void_p, f32p, C.c_size_t]
    L.rt_renderer_shard_floats.argtypes = [C.c_void_p, P(C.c_size_t)]
    L.rt_renderer_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.rt_probe_aabb.argtypes = [C.c_int, C.c_size_t, f32p, f32p, f32p, i32p, f32p]
    L.rt_probe_sphere.argtypes = [C.c_int, C.c_size_t, f32p, f32p, f32p]
    L.rt_probe_trace.argtypes = [C.c_int, P(WorldFlat), C.c_size_t, f32p, i32p, f32p, i32p, f32p]
    L.rt_probe_scatter.argtypes = [C.c_int, C.c_uint64, C.c_size_t, C.c_void_p, f32p, f32p, f32p, u32p, i32p, f32p, f32p, u32p]
    L.rt_probe_camera.argtypes = [C.c_int, C.c_uint64, P(Camera), C.c_size_t, f32p, u32p, f32p, u32p]
    L.rt_probe_radiance.argtypes = [P(RenderConfig), P(Camera), P(WorldFlat), C.c_size_t, u32p, f32p]
    L.rt_probe_sphere_index.argtypes = [C.c_int, P(Camera), C.c_uint32, C.c_uint32, C.c_size_t, f32p, i32p]
    L.rt_probe_rng.argtypes = [C.c_int, C.c_uint64, C.c_size_t, u32p, C.c_uint32, f32p]
    L.rt_probe_math.argtypes = [C.c_int32, C.c_int32, C.c_size_t, f32p, f32p, f32p]
    L.rt_probe_glm.argtypes = [C.c_int32, C.c_int32, C.c_size_t, f32p, f32p]
    L.rt_probe_aabb_misc.argtypes = [C.c_size_t, f32p, f32p]
    L.rt_probe_aabb_regular.argtypes = [C.c_int, C.c_size_t, f32p, f32p, f32p, i32p, i32p, f32p]
    L.rt_probe_boxpair_filtered.argtypes = [C.c_int, C.c_size_t, f32p, f32p, f32p, i32p]
    L.rt_probe_boxpair_certified.argtypes = [C.c_int, C.c_size_t, f32p, f32p, f32p, i32p]
    L.rt_selftest_fastdiv.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, P(C.c_uint64), u32p]
    L.rt_selftest_fastdiv4.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, P(C.c_uint64), u32p]
    L.rt_selftest_fastrcp.argtypes = [C.c_int32, P(C.c_uint64), P(C.c_uint64), P(C.c_uint32)]
    L.rt_device_count.argtypes = [P(C.c_int)]
    _lib = L
    return L


def check(rc):
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error().decode("utf-8", "replace"))


def v3(a):
    return vec3(*[float(x) for x in a])
