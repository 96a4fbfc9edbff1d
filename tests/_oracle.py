"""ctypes binding of the CPU oracle (oracle/liboracle.so) — test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


class Node(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("max", C.c_float * 3), ("left", C.c_int32), ("right", C.c_int32)]


class Prim(C.Structure):
    _fields_ = [("c0", C.c_float * 3), ("radius", C.c_float), ("c1", C.c_float * 3), ("mat", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("param", C.c_float), ("albedo2", C.c_float * 3), ("type", C.c_uint32)]


class World(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("root", C.c_int32), ("n_nodes", C.c_uint32), ("n_prims", C.c_uint32),
                ("n_materials", C.c_uint32), ("max_stack", C.c_uint32),
                ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3),
                ("nodes", C.c_void_p), ("prims", C.c_void_p), ("materials", C.c_void_p),
                ("quads", C.c_void_p), ("n_quads", C.c_uint32), ("background", C.c_uint32),
                ("background_color", C.c_float * 3), ("image_width", C.c_uint32),
                ("perlin", C.c_void_p), ("image", C.c_void_p), ("image_height", C.c_uint32), ("traversal", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("type", C.c_uint32), ("o", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
                ("w", C.c_float * 3), ("viewport_width", C.c_float), ("viewport_height", C.c_float),
                ("lens_radius", C.c_float), ("focus_dist", C.c_float), ("t0", C.c_float), ("t1", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("box_tests", C.c_uint64),
                ("leaf_tests", C.c_uint64), ("shaded_hits", C.c_uint64), ("rng_draws", C.c_uint64),
                ("max_stack", C.c_uint32)]


NODE_DT = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("left", "<i4"), ("right", "<i4")])
PRIM_DT = np.dtype([("c0", "<f4", 3), ("radius", "<f4"), ("c1", "<f4", 3), ("mat", "<u4")])
MAT_DT = np.dtype([("albedo", "<f4", 3), ("param", "<f4"), ("albedo2", "<f4", 3), ("type", "<u4")])
QUAD_DT = np.dtype([("Q", "<f4", 3), ("D", "<f4"), ("u", "<f4", 3), ("mat", "<u4"), ("v", "<f4", 3), ("pad0", "<f4"),
                    ("normal", "<f4", 3), ("pad1", "<f4"), ("w", "<f4", 3), ("pad2", "<f4")])

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(ORACLE_DIR, "rt_oracle.c")):
        build()
    L = C.CDLL(path)
    v3 = C.c_float * 3
    L.orc_glm_dot.restype = C.c_float
    L.orc_glm_mix1.restype = C.c_float
    L.orc_glm_mix1.argtypes = [C.c_float] * 3
    L.orc_glm_compmax.restype = C.c_float
    L.orc_glm_compmin.restype = C.c_float
    L.orc_glm_length2.restype = C.c_float
    L.orc_glm_radians.restype = C.c_float
    L.orc_glm_radians.argtypes = [C.c_float]
    L.orc_glm_refract.argtypes = [v3, v3, C.c_float, v3]
    L.orc_glm_mix3.argtypes = [v3, v3, C.c_float, v3]
    L.orc_glm_lerp.argtypes = [v3, v3, C.c_float, v3]
    L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
    L.orc_rng_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p]
    L.orc_math_batch.argtypes = [C.c_int, C.c_size_t, f32p, f32p, f32p]
    L.orc_scene_set_perlin.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_scene_book2_final.argtypes = [C.c_uint64]
    L.orc_scene_book2_final.restype = C.c_void_p
    L.orc_scene_set_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.orc_aabb_batch.argtypes = [C.c_size_t, f32p, f32p, f32p, i32p, f32p]
    L.orc_sphere_batch.argtypes = [C.c_size_t, f32p, f32p, f32p]
    L.orc_trace_batch.argtypes = [C.POINTER(World), C.c_size_t, f32p, i32p, f32p, i32p, f32p]
    L.orc_scatter_batch.argtypes = [C.c_uint64, C.c_size_t, C.c_void_p, f32p, f32p, f32p, u32p, i32p, f32p, f32p, u32p]
    L.orc_camera_batch.argtypes = [C.c_uint64, C.POINTER(Camera), C.c_size_t, f32p, u32p, f32p, u32p]
    L.orc_radiance_batch.argtypes = [C.POINTER(World), C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_uint64, C.c_size_t, u32p, f32p]
    L.orc_sphere_index.argtypes = [C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_size_t, f32p, i32p]
    for name in ("orc_camera_pinhole",):
        getattr(L, name).argtypes = [v3, v3, v3, C.c_float, C.c_float, C.POINTER(Camera)]
    L.orc_camera_defocus.argtypes = [v3, v3, v3, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Camera)]
    L.orc_camera_motion.argtypes = [v3, v3, v3, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Camera)]
    L.orc_render.argtypes = [C.POINTER(World), C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                             C.c_uint64, C.c_int, f32p, C.POINTER(Counters)]
    L.orc_render.restype = C.c_int
    L.orc_render_pixels.argtypes = [C.POINTER(World), C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint64, C.c_size_t, u32p, f32p]
    L.orc_render_pixels.restype = C.c_int
    for name in ("orc_scene_book1_final", "orc_scene_book2_moving"):
        getattr(L, name).argtypes = [C.c_uint64]
        getattr(L, name).restype = C.c_void_p
    L.orc_scene_three_spheres.restype = C.c_void_p
    L.orc_scene_cornell_box.restype = C.c_void_p
    L.orc_scene_from_arrays_ext.argtypes = [C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    L.orc_scene_from_arrays_ext.restype = C.c_void_p
    L.orc_scene_from_arrays.argtypes = [C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    L.orc_scene_from_arrays.restype = C.c_void_p
    L.orc_scene_world.argtypes = [C.c_void_p, C.POINTER(World)]
    L.orc_scene_free.argtypes = [C.c_void_p]
    _lib = L
    return L


def vec3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


class Scene:
    """Owns an orc_scene; exposes the flat world and numpy views of its arrays."""

    def __init__(self, handle):
        self.h = handle
        self.world = World()
        lib().orc_scene_world(self.h, C.byref(self.world))

    @classmethod
    def book1_final(cls, seed=1984):
        return cls(lib().orc_scene_book1_final(seed))

    @classmethod
    def book2_moving(cls, seed=1984):
        return cls(lib().orc_scene_book2_moving(seed))

    @classmethod
    def three_spheres(cls):
        return cls(lib().orc_scene_three_spheres())

    @classmethod
    def book2_final(cls, seed=1984):
        return cls(lib().orc_scene_book2_final(seed))

    @classmethod
    def cornell_box(cls):
        return cls(lib().orc_scene_cornell_box())

    @classmethod
    def from_arrays_ext(cls, prims, quads, mats, builder=0, background=0, background_color=(0, 0, 0)):
        prims = np.ascontiguousarray(prims, dtype=PRIM_DT)
        quads = np.ascontiguousarray(quads, dtype=QUAD_DT)
        mats = np.ascontiguousarray(mats, dtype=MAT_DT)
        bg = (C.c_float * 3)(*[float(x) for x in background_color])
        return cls(lib().orc_scene_from_arrays_ext(len(prims), prims.ctypes.data if len(prims) else None, len(quads),
                                                   quads.ctypes.data if len(quads) else None, len(mats), mats.ctypes.data,
                                                   builder, background, bg))

    @classmethod
    def from_arrays(cls, prims, mats, builder=0):
        prims = np.ascontiguousarray(prims, dtype=PRIM_DT)
        mats = np.ascontiguousarray(mats, dtype=MAT_DT)
        return cls(lib().orc_scene_from_arrays(len(prims), prims.ctypes.data, len(mats), mats.ctypes.data, builder))

    def _arr(self, ptr, n, dt):
        if n == 0 or not ptr:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).copy()

    @property
    def nodes(self):
        return self._arr(self.world.nodes, self.world.n_nodes, NODE_DT)

    @property
    def prims(self):
        return self._arr(self.world.prims, self.world.n_prims, PRIM_DT)

    @property
    def quads(self):
        return self._arr(self.world.quads, self.world.n_quads, QUAD_DT)

    def set_perlin(self, seed=1984):
        lib().orc_scene_set_perlin(self.h, seed)
        lib().orc_scene_world(self.h, C.byref(self.world))
        return self

    def set_image(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        lib().orc_scene_set_image(self.h, rgb.shape[1], rgb.shape[0], rgb.ctypes.data)
        lib().orc_scene_world(self.h, C.byref(self.world))
        return self

    @property
    def perlin(self):
        buf = (C.c_char * 6144).from_address(self.world.perlin)
        return bytes(buf)

    @property
    def materials(self):
        return self._arr(self.world.materials, self.world.n_materials, MAT_DT)

    def __del__(self):
        try:
            lib().orc_scene_free(self.h)
        except Exception:
            pass


def camera_pinhole(lookfrom, lookat, up, vfov, aspect):
    c = Camera()
    lib().orc_camera_pinhole(vec3(lookfrom), vec3(lookat), vec3(up), vfov, aspect, C.byref(c))
    return c


def camera_defocus(lookfrom, lookat, up, vfov, aspect, aperture, focus_dist):
    c = Camera()
    lib().orc_camera_defocus(vec3(lookfrom), vec3(lookat), vec3(up), vfov, aspect, aperture, focus_dist, C.byref(c))
    return c


def camera_motion(lookfrom, lookat, up, vfov, aspect, t0, t1):
    c = Camera()
    lib().orc_camera_motion(vec3(lookfrom), vec3(lookat), vec3(up), vfov, aspect, t0, t1, C.byref(c))
    return c


def render(world, cam, width, height, spp, max_depth, seed=1984, threads=None):
    if threads is None:
        threads = os.cpu_count() or 1
    out = np.zeros((height, width, 4), dtype=np.float32)
    cnt = Counters()
    rc = lib().orc_render(C.byref(world), C.byref(cam), width, height, spp, max_depth, seed, threads, out, C.byref(cnt))
    if rc != 0:
        raise RuntimeError(f"orc_render failed rc={rc}")
    return out, cnt


def render_pixels(world, cam, width, height, spp, max_depth, gids, seed=1984):
    gids = np.ascontiguousarray(gids, dtype=np.uint32)
    out = np.zeros((len(gids), 4), dtype=np.float32)
    rc = lib().orc_render_pixels(C.byref(world), C.byref(cam), width, height, spp, max_depth, seed, len(gids), gids, out)
    if rc != 0:
        raise RuntimeError(f"orc_render_pixels failed rc={rc}")
    return out
