"""Python view of the reference-shaped host API, over the C ABI (used by tests/ and bench.py).

Names follow the reference: Sphere / MovingSphere, material descriptors, SphereHandle-style adders,
BVH_Handle.Factory builders, the three cameras, and Renderer.MakeRenderer / Render /
DownloadRenderbuffer (main/src/Renderer.h:38-46).  The C++ twin of this file is include/rt06/*.hpp.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import Camera, RenderConfig, WorldFlat, check, lib, v3


def PinholeCamera(lookfrom, lookat, up, vfov, aspect_ratio):
    """PinholeCamera ctor, rt_engine/shaders/cu_Cameras.cuh:16-25."""
    c = Camera()
    check(lib().rt_camera_pinhole(v3(lookfrom), v3(lookat), v3(up), vfov, aspect_ratio, C.byref(c)))
    return c


def DefocusBlurCamera(lookfrom, lookat, up, vfov, aspect_ratio, aperture, focus_dist):
    """DefocusBlurCamera ctor, cu_Cameras.cuh:40-52."""
    c = Camera()
    check(lib().rt_camera_defocus(v3(lookfrom), v3(lookat), v3(up), vfov, aspect_ratio, aperture, focus_dist, C.byref(c)))
    return c


def MotionBlurCamera(lookfrom, lookat, up, vfov, aspect_ratio, time0, time1):
    """MotionBlurCamera ctor, cu_Cameras.cuh:73-85."""
    c = Camera()
    check(lib().rt_camera_motion(v3(lookfrom), v3(lookat), v3(up), vfov, aspect_ratio, time0, time1, C.byref(c)))
    return c


class Scene:
    """Host scene: what SphereHandle / newOnDevice<Material> / BVH_Handle::Factory / HittableList /
    bvh_node build in the reference, kept as flat arrays."""

    def __init__(self, handle=None):
        if handle is None:
            h = C.c_void_p()
            check(lib().rt_scene_create(C.byref(h)))
            handle = h
        self.h = handle

    # --- prefab scenes (SceneBook2BVH::Factory::MakeScene and friends) ---
    @classmethod
    def book1_final(cls, seed=1984):
        h = C.c_void_p()
        check(lib().rt_scene_book1_final(seed, C.byref(h)))
        return cls(h)

    @classmethod
    def book2_moving(cls, seed=1984):
        h = C.c_void_p()
        check(lib().rt_scene_book2_moving(seed, C.byref(h)))
        return cls(h)

    @classmethod
    def three_spheres(cls):
        h = C.c_void_p()
        check(lib().rt_scene_three_spheres(C.byref(h)))
        return cls(h)

    @classmethod
    def cornell_box(cls):
        """BASELINE.json configs[3] (not in the reference): 18 quads, one light, black background."""
        h = C.c_void_p()
        check(lib().rt_scene_cornell_box(C.byref(h)))
        return cls(h)

    @classmethod
    def book2_final(cls, seed=1984):
        """BASELINE.json configs[4] (not in the reference): final_scene() of "The Next Week"."""
        h = C.c_void_p()
        check(lib().rt_scene_book2_final(seed, C.byref(h)))
        return cls(h)

    # --- vocabulary ---
    def add_material(self, mtype, albedo, param=0.0, albedo2=None):
        out = C.c_int32()
        a2 = v3(albedo2) if albedo2 is not None else None
        check(lib().rt_scene_add_material(self.h, mtype, v3(albedo), param, a2, C.byref(out)))
        return out.value

    def Lambertian(self, albedo):
        return self.add_material(capi.MAT_LAMBERTIAN, albedo)

    def Metal(self, albedo, fuzz):
        return self.add_material(capi.MAT_METAL, albedo, fuzz)

    def Dielectric(self, albedo, ior):
        return self.add_material(capi.MAT_DIELECTRIC, albedo, ior)

    def LambertianTexture(self, c1, c2, scale):
        return self.add_material(capi.MAT_LAMBERTIAN_CHECKER, c1, np.float32(1.0) / np.float32(scale), c2)

    def DiffuseLight(self, emit):
        """diffuse_light of "The Next Week" (extension, not in the reference)."""
        return self.add_material(capi.MAT_DIFFUSE_LIGHT, emit)

    def Isotropic(self, albedo, density):
        """isotropic phase function of "The Next Week" (extension): a sphere made of it is a constant_medium of that density."""
        return self.add_material(capi.MAT_ISOTROPIC, albedo, density)

    def MakeConstantMedium(self, center, radius, density, albedo):
        """constant_medium(sphere(center, radius), density, albedo) of "The Next Week" (extension, not in the reference)."""
        return self.MakeSphere(center, radius, self.Isotropic(albedo, density))

    def set_perlin(self, seed=1984):
        """perlin::perlin() of "The Next Week" (extension): the world's noise tables."""
        check(lib().rt_scene_set_perlin(self.h, seed))
        return self

    def NoiseTexture(self, scale, albedo=(0.5, 0.5, 0.5)):
        """lambertian(noise_texture(scale)) of "The Next Week" (extension); needs set_perlin()."""
        return self.add_material(capi.MAT_LAMBERTIAN_NOISE, albedo, scale)

    def set_image(self, rgb):
        """The image of image_texture (extension): uint8 array [H][W][3], row 0 = top."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        check(lib().rt_scene_set_image(self.h, rgb.shape[1], rgb.shape[0], rgb.ctypes.data))
        return self

    def ImageTexture(self):
        """lambertian(image_texture) of "The Next Week" (extension) on spheres (u, v from the normal) and quads (u, v = the planar coordinates of the hit); needs set_image()."""
        return self.add_material(capi.MAT_LAMBERTIAN_IMAGE, (1, 1, 1))

    def MakeBox(self, a, b, mat, rotate_y=0.0, translate=(0, 0, 0)):
        """box(a, b, mat) of "The Next Week" as 6 quads, optionally rotate_y(degrees) then translate; returns the first quad index."""
        out = C.c_int32()
        check(lib().rt_scene_add_box(self.h, v3(a), v3(b), mat, rotate_y, v3(translate), C.byref(out)))
        return out.value

    def MakeQuad(self, Q, u, v, mat):
        """quad(Q,u,v,mat) of "The Next Week" (extension, not in the reference)."""
        out = C.c_int32()
        check(lib().rt_scene_add_quad(self.h, v3(Q), v3(u), v3(v), mat, C.byref(out)))
        return out.value

    def set_background(self, color=None):
        """None: the reference's sky gradient; a colour: camera::background of "The Next Week"."""
        if color is None:
            check(lib().rt_scene_set_background(self.h, 0, v3((0, 0, 0))))
        else:
            check(lib().rt_scene_set_background(self.h, 1, v3(color)))
        return self

    def MakeSphere(self, center, radius, mat):
        out = C.c_int32()
        check(lib().rt_scene_add_sphere(self.h, v3(center), radius, mat, C.byref(out)))
        return out.value

    def MakeMovingSphere(self, c0, c1, radius, mat):
        out = C.c_int32()
        check(lib().rt_scene_add_moving_sphere(self.h, v3(c0), v3(c1), radius, mat, C.byref(out)))
        return out.value

    def prim_bounds(self, prim):
        mn, mx = capi.vec3(), capi.vec3()
        check(lib().rt_scene_prim_bounds(self.h, prim, mn, mx))
        return np.array(mn[:], dtype=np.float32), np.array(mx[:], dtype=np.float32)

    def BuildBVH_TopDown(self):
        check(lib().rt_scene_build_bvh_topdown(self.h))
        return self

    def BuildBVH_SAH(self):
        check(lib().rt_scene_build_bvh_sah(self.h))
        return self

    def BuildBVH_BottomUp(self):
        check(lib().rt_scene_build_bvh_bottomup(self.h))
        return self

    def MakeHittableList(self):
        check(lib().rt_scene_set_world_list(self.h))
        return self

    def bvh_node(self, left_ref, right_ref, bounds=None):
        out = C.c_int32()
        if bounds is None:
            check(lib().rt_scene_add_bvh_node(self.h, left_ref, right_ref, None, None, C.byref(out)))
        else:
            mn, mx = v3(bounds[0]), v3(bounds[1])
            check(lib().rt_scene_add_bvh_node(self.h, left_ref, right_ref, mn, mx, C.byref(out)))
        return out.value

    @staticmethod
    def prim_ref(prim):
        return -prim - 1

    def set_world_node_tree(self, root_ref):
        check(lib().rt_scene_set_world_node_tree(self.h, root_ref))
        return self

    def set_traversal(self, mode):
        """0 = the live depth-first stack (BVH.cu:54-106), 1 = the reference's disabled distance-sorted queue (BVH.cu:17-49), fixed,
        2 = a 4-wide walk of the same tree (two levels per visit; not in the reference)"""
        check(lib().rt_scene_set_traversal(self.h, mode))
        return self

    # --- flat view ---
    def getWorldPtr(self):
        w = WorldFlat()
        check(lib().rt_scene_get_flat(self.h, C.byref(w)))
        return w

    def _arr(self, ptr, n, dt):
        if n == 0 or not ptr:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).copy()

    def arrays(self):
        w = self.getWorldPtr()
        return (self._arr(w.nodes, w.n_nodes, capi.NODE_DT), self._arr(w.prims, w.n_prims, capi.PRIM_DT),
                self._arr(w.materials, w.n_materials, capi.MAT_DT))

    def quads(self):
        w = self.getWorldPtr()
        return self._arr(w.quads, w.n_quads, capi.QUAD_DT)

    def perlin_bytes(self):
        w = self.getWorldPtr()
        return bytes((C.c_char * 6144).from_address(w.perlin)) if w.perlin else b""

    def image(self):
        w = self.getWorldPtr()
        if not w.image:
            return np.zeros((0, 0, 3), np.uint8)
        buf = (C.c_char * (w.image_width * w.image_height * 3)).from_address(w.image)
        return np.frombuffer(buf, dtype=np.uint8).reshape(w.image_height, w.image_width, 3).copy()

    def __del__(self):
        try:
            if self.h:
                lib().rt_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Renderer:
    """Renderer (main/src/Renderer.h:12-47) over the C ABI."""

    def __init__(self, handle, cfg):
        self.h = handle
        self.cfg = cfg

    @classmethod
    def MakeRenderer(cls, render_width, render_height, samples_per_pixel, max_depth, cam, world,
                     seed=1984, device=0, rank=0, world_size=1, variant=0):
        cfg = RenderConfig(render_width, render_height, samples_per_pixel, max_depth, seed, device, rank, world_size, variant)
        h = C.c_void_p()
        check(lib().rt_renderer_create(C.byref(cfg), C.byref(cam), C.byref(world), C.byref(h)))
        return cls(h, cfg)

    def Render(self):
        check(lib().rt_renderer_render(self.h))

    def render_async(self, stream=None, d_out=None):
        check(lib().rt_renderer_render_async(self.h, C.c_void_p(stream or 0), C.c_void_p(d_out or 0)))

    def last_kernel_ms(self):
        ms = C.c_float()
        check(lib().rt_renderer_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def kernel_times(self, renders_back=0):
        """(primary_rays_kernel, dominant kernel, resolve_kernel) ms of one of the last 32 render calls, by HIP events, summed over its passes."""
        out = (C.c_float * 3)()
        check(lib().rt_renderer_kernel_times(self.h, renders_back, out))
        return tuple(out)

    def pass_info(self):
        """{'n_passes', 'pass_spp', 'bytes_per_sample', 'buffer_bytes'}: how a render is cut into passes (rt_renderer_pass_info)."""
        out = (C.c_uint64 * 4)()
        check(lib().rt_renderer_pass_info(self.h, out))
        return {"n_passes": out[0], "pass_spp": out[1], "bytes_per_sample": out[2], "buffer_bytes": out[3]}

    def kernel_info(self):
        """{'variant', 'lds_resident', 'workgroup', 'workgroups_per_cu'} the renderer resolved to."""
        out = (C.c_uint32 * 4)()
        check(lib().rt_renderer_kernel_info(self.h, C.byref(out)))
        return {"variant": out[0], "lds_resident": bool(out[1]), "workgroup": out[2], "workgroups_per_cu": out[3]}

    def DownloadRenderbuffer(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 4), dtype=np.float32)
        check(lib().rt_renderer_download(self.h, out, out.size))
        return out

    def shard_floats(self):
        n = C.c_size_t()
        check(lib().rt_renderer_shard_floats(self.h, C.byref(n)))
        return n.value

    def assemble(self, d_gathered, d_image, stream=None):
        check(lib().rt_renderer_assemble(self.h, C.c_void_p(d_gathered), C.c_void_p(d_image), C.c_void_p(stream or 0)))

    def close(self):
        if self.h:
            lib().rt_renderer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiRenderer:
    """Renderer over the N GPUs of one node from ONE process (rt_multi_renderer_*): tile shards, one RCCL gather at frame end."""

    def __init__(self, handle, cfg):
        self.h, self.cfg = handle, cfg

    @classmethod
    def MakeRenderer(cls, render_width, render_height, samples_per_pixel, max_depth, cam, world, n_gpus, seed=1984, devices=None, variant=0):
        cfg = RenderConfig(render_width, render_height, samples_per_pixel, max_depth, seed, 0, 0, 1, variant)
        devs = (C.c_int32 * n_gpus)(*devices) if devices is not None else None
        h = C.c_void_p()
        check(lib().rt_multi_renderer_create(C.byref(cfg), C.byref(cam), C.byref(world), n_gpus, devs, C.byref(h)))
        return cls(h, cfg)

    def Render(self):
        check(lib().rt_multi_renderer_render(self.h))

    def DownloadRenderbuffer(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 4), dtype=np.float32)
        check(lib().rt_multi_renderer_download(self.h, out, out.size))
        return out

    def times(self):
        """(host wall-clock of Render, slowest rank's kernels, exchange + assembly on GPU 0) in ms"""
        out = (C.c_float * 3)()
        check(lib().rt_multi_renderer_times(self.h, out))
        return tuple(out)

    def close(self):
        if self.h:
            lib().rt_multi_renderer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_info(device=0):
    """rt_device_info: {'compute_units', 'clock_khz', 'memory_mib', 'memory_clock_khz'}"""
    out = (C.c_uint32 * 4)()
    check(lib().rt_device_info(device, out))
    return {"compute_units": out[0], "clock_khz": out[1], "memory_mib": out[2], "memory_clock_khz": out[3]}


def shard_layout(width, height, world_size):
    """rt_shard_layout (host): (tiles_x, n_tiles, n_local_tiles, shard_floats)"""
    out = (C.c_uint32 * 4)()
    check(lib().rt_shard_layout(width, height, world_size, out))
    return tuple(out)


def shard_pixel_map(width, height, world_size, rank):
    """rt_shard_pixel_map (host): global pixel id of every position of rank's shard, 0xffffffff for padding"""
    n = shard_layout(width, height, world_size)[2] * 64
    out = np.zeros(n, np.uint32)
    check(lib().rt_shard_pixel_map(width, height, world_size, rank, out, n))
    return out


# --- device probes -------------------------------------------------------------------------------
def probe_aabb(boxes, rays, max_dist, device=0):
    n = len(boxes)
    hit = np.zeros(n, np.int32); dist = np.zeros(n, np.float32)
    check(lib().rt_probe_aabb(device, n, np.ascontiguousarray(boxes, np.float32), np.ascontiguousarray(rays, np.float32),
                              np.ascontiguousarray(max_dist, np.float32), hit, dist))
    return hit, dist


def probe_sphere(rays, spheres, device=0):
    n = len(rays)
    t = np.zeros(n, np.float32)
    check(lib().rt_probe_sphere(device, n, np.ascontiguousarray(rays, np.float32), np.ascontiguousarray(spheres, np.float32), t))
    return t


def probe_trace(world, rays, device=0):
    n = len(rays)
    hit = np.zeros(n, np.int32); t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); nrm = np.zeros((n, 3), np.float32)
    check(lib().rt_probe_trace(device, C.byref(world), n, np.ascontiguousarray(rays, np.float32), hit, t, prim, nrm))
    return hit, t, prim, nrm


def probe_scatter(seed, mats, rays, dist, normals, keys, device=0):
    n = len(rays)
    mats = np.ascontiguousarray(mats, dtype=capi.MAT_DT)
    sc = np.zeros(n, np.int32); orays = np.zeros((n, 7), np.float32); att = np.zeros((n, 3), np.float32); draws = np.zeros(n, np.uint32)
    check(lib().rt_probe_scatter(device, seed, n, mats.ctypes.data, np.ascontiguousarray(rays, np.float32),
                                 np.ascontiguousarray(dist, np.float32), np.ascontiguousarray(normals, np.float32),
                                 np.ascontiguousarray(keys, np.uint32), sc, orays, att, draws))
    return sc, orays, att, draws


def probe_camera(seed, cam, st, keys, device=0):
    n = len(st)
    orays = np.zeros((n, 7), np.float32); draws = np.zeros(n, np.uint32)
    check(lib().rt_probe_camera(device, seed, C.byref(cam), n, np.ascontiguousarray(st, np.float32),
                                np.ascontiguousarray(keys, np.uint32), orays, draws))
    return orays, draws


def probe_radiance(cfg, cam, world, keys):
    n = len(keys)
    out = np.zeros((n, 3), np.float32)
    check(lib().rt_probe_radiance(C.byref(cfg), C.byref(cam), C.byref(world), n, np.ascontiguousarray(keys, np.uint32), out))
    return out


def probe_sphere_index(cam, width, height, spheres, device=0):
    out = np.zeros(width * height, np.int32)
    spheres = np.ascontiguousarray(spheres, np.float32)
    check(lib().rt_probe_sphere_index(device, C.byref(cam), width, height, len(spheres), spheres, out))
    return out.reshape(height, width)


def probe_rng(seed, keys, n_draws, device=0):
    n = len(keys)
    out = np.zeros((n, n_draws), np.float32)
    check(lib().rt_probe_rng(device, seed, n, np.ascontiguousarray(keys, np.uint32), n_draws, out))
    return out


def probe_math(fn, a, b=None, device=0):
    """rt_probe_math: fn 0 log, 1 sin, 2 acos, 3 atan2(a, b)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(a if b is None else b, np.float32)
    out = np.zeros_like(a)
    check(lib().rt_probe_math(device, fn, len(a), a, b, out))
    return out


GLM_FUNCTIONS = ("dot", "cross", "normalize", "reflect", "refract", "mix3", "mix1", "min3", "max3", "compmax", "compmin",
                 "clamp01_sqrt", "near_zero", "length2", "lerp", "radians", "ray")
_GLM_SHAPES = ((6, 1), (6, 3), (3, 3), (6, 3), (7, 3), (7, 3), (3, 1), (6, 3), (6, 3), (3, 1), (3, 1), (3, 3), (3, 1), (3, 1), (7, 3), (1, 1), (10, 4))


def probe_glm(name, inputs, device=0):
    """rt_probe_glm: one function of the device math vocabulary (GLM_FUNCTIONS) over an (n, nin) array -> (n, nout)."""
    fn = GLM_FUNCTIONS.index(name)
    nin, nout = _GLM_SHAPES[fn]
    a = np.ascontiguousarray(inputs, np.float32).reshape(-1, nin)
    out = np.zeros((len(a), nout), np.float32)
    check(lib().rt_probe_glm(device, fn, len(a), a, out))
    return out


def probe_aabb_misc(boxes):
    """rt_probe_aabb_misc (host): (n, 12) boxes a, b -> (n, 20) [axis, area, centroid, union, a += b, compares]."""
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 12)
    out = np.zeros((len(b), 20), np.float32)
    check(lib().rt_probe_aabb_misc(len(b), b, out))
    return out


def device_count():
    n = C.c_int()
    check(lib().rt_device_count(C.byref(n)))
    return n.value


def probe_aabb_regular(boxes, rays, max_dist, device=0):
    n = len(boxes)
    reg = np.zeros(n, np.int32); hit = np.zeros(n, np.int32); dist = np.zeros(n, np.float32)
    check(lib().rt_probe_aabb_regular(device, n, np.ascontiguousarray(boxes, np.float32), np.ascontiguousarray(rays, np.float32),
                                      np.ascontiguousarray(max_dist, np.float32), reg, hit, dist))
    return reg, hit, dist


def selftest_fastdiv(first_den, n_den, num_exp=0, den_exp=0, device=0, four=False):
    """(mismatching pairs, example) over n_den divisor significands x all 2^23 numerator significands;
    four=True checks the 4-instruction two-word-reciprocal form (fast_div_exact4)."""
    bad = C.c_uint64()
    ex = np.zeros(2, np.uint32)
    fn = lib().rt_selftest_fastdiv4 if four else lib().rt_selftest_fastdiv
    check(fn(device, first_den, n_den, num_exp, den_exp, C.byref(bad), ex))
    return bad.value, ex


def selftest_fastrcp(device=0):
    """rt_selftest_fastrcp: (values checked, mismatches, example bits)."""
    n, bad, ex = C.c_uint64(), C.c_uint64(), C.c_uint32()
    check(lib().rt_selftest_fastrcp(device, C.byref(n), C.byref(bad), C.byref(ex)))
    return n.value, bad.value, ex.value


def probe_boxpair_certified(boxes, rays, max_dist, device=0):
    """rt_probe_boxpair_certified: the default hot loop's box pair next to the verbatim box tests, (n, 8) int32."""
    n = len(boxes)
    out = np.zeros((n, 8), np.int32)
    check(lib().rt_probe_boxpair_certified(device, n, np.ascontiguousarray(boxes, np.float32), np.ascontiguousarray(rays, np.float32),
                                           np.ascontiguousarray(max_dist, np.float32), out))
    return out


def probe_boxpair_filtered(boxes, rays, max_dist, device=0):
    n = len(boxes)
    out = np.zeros((n, 8), np.int32)
    check(lib().rt_probe_boxpair_filtered(device, n, np.ascontiguousarray(boxes, np.float32), np.ascontiguousarray(rays, np.float32),
                                          np.ascontiguousarray(max_dist, np.float32), out))
    return out
