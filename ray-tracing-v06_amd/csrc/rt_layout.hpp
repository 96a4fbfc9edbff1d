// rt_layout.hpp — what the host runtime and the kernels agree on: pixel ownership (TileMap), the packed scene image of the streaming kernels
// (wide nodes, reference encoding, PackedSceneRef) and the sizes the ray-exchange kernel's error buffer is laid out with.  No kernels here: every
// translation unit of librt06.so may include it.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "rt06.h"
#include "rt_internal.hpp"

// Pixel ownership.  The frame is cut into 8x8 tiles (row-major tile order); tile t belongs to rank
// t % world_size.  A rank's pixels are enumerated tile-major: L = local_tile * 64 + (py * 8 + px).
struct TileMap {
    uint32_t width, height, tiles_x, n_tiles;
    uint32_t rank, world_size, n_local_tiles;
    uint32_t direct;  // 1: write row-major at gid (single GPU); 0: write the compact shard at L
};

// Wide node, 19 dwords (76 B): both child boxes + both child references.
//   dwords 3*(3*side + axis) .. +2 = (min, max, min) of that axis of the left (side 0) / right (side 1) child box;
//   dword 18 = left reference | right reference << 16.
// The (min, max, min) triple lets a lane read its (near, far) plane pair of an axis as two CONSECUTIVE dwords at
// offset 0 (direction >= 0: near = min) or 4 bytes (direction < 0: near = max), i.e. the per-axis min/max of the
// slab test (aabb.cuh:34-39) becomes a per-ray address offset instead of 12 v_min/v_max per visit — on gfx950 min,
// max, compares and selects issue at half the rate of add/mul/fma (tools/bench_valu_issue.hip).  The odd stride
// also spreads the rows of different nodes over all LDS banks.
// A reference is 16 bits wide (an LDS-resident scene has < 2^15 inner nodes and leaf codes):
//   bit 15 clear: wide-node index;  bit 15 set: leaf, code = ref & 0x7fff = prim * 2 + is_moving.
// The same encoding travels through the per-lane LDS stack as 16-bit entries.
#define RT_REF_LEAF 0x8000u
#define RT_REF_IRR 0x4000u   // see FAST_BVH in render_kernel_stream
#define RT_MAT_INDEX_MASK 0x0fffffffu  // matbits: index (28 bits) | moving << 28 | type << 29
#define RT_NODE_DWORDS 19u
#define RT_NODE_BYTES (RT_NODE_DWORDS * 4u)
#define RT_NODE_REFS 18u
// BIG scenes (the image does not fit the LDS, or has 2^14 inner nodes / 2^15 leaf codes or more): the records are read
// from global memory (they stay L2-resident) and references are 32 bits wide (bit 31 marks a leaf, bit 30 a ray outside
// the fast class); the per-lane stacks (32-bit entries) are all that lives in the LDS.  That path is bound by the L1's
// tag-lookup rate — 64 lanes reading 64 different nodes cost ~20 lookups per load instruction (PMC) — not by the vector
// issue, so its wide node is ONE 64-byte line read with four 16-byte loads, [lmin.xyz lmax.x | lmax.yz rmin.xy |
// rmin.z rmax.xyz | left, right, -, -], and near / far planes are picked with selects instead of by address.
#define RT_REF_LEAF_BIG 0x80000000u
#define RT_REF_IRR_BIG 0x40000000u
#define RT_NODE_DWORDS_BIG 16u
// number of 16-B units the node region of `n` wide nodes occupies in the blob
#define RT_NODES_VEC4(n, big) (((n) * ((big) ? RT_NODE_DWORDS_BIG : RT_NODE_DWORDS) + 3u) / 4u)

// LDS image, in 16-B units:  [wide nodes (RT_NODE_DWORDS dwords each, region rounded up) | spheres (c0, r) | extra (c1, matbits) | mats16 (albedo, param) |
//                              (padding to a 64-byte boundary) quads (4 each: Q,D | u,v.x | v.yz,n.xy | n.z,w) | quad shade records (normal, matbits)],   matbits = material index | moving << 28 | type << 29
struct PackedSceneRef {
    const uint4* blob;
    uint32_t blob_vec4;      // number of 16-B units to stage into LDS
    uint32_t off_spheres;
    uint32_t off_extra;
    uint32_t off_mats;
    uint32_t off_quads;
    uint32_t sphere_codes;   // leaf codes below this are spheres (prim * 2 + is_moving); code - sphere_codes is a quad index
    uint32_t background;     // 0 = reference sky gradient, 1 = background_color
    float background_color[3];
    uint32_t root_ref;
    float root_min[3], root_max[3];
    uint32_t stack_cap;      // entries per lane
    uint32_t n_inner, n_codes, n_prims, n_quads;
    uint32_t n_top;          // BIG: the first n_top wide nodes (breadth-first order = the top of the tree) are staged in the LDS
    const rt_material* mats; // full 32-B records in global memory (second colour of a checker material)
    const rt_perlin* perlin; // EXT: noise tables / image of the two textured materials (global memory), or null
    const uint8_t* image;
    uint32_t image_w, image_h;
};

#define RT_STREAM_BLOCK 768      // default workgroup: 12 wavefronts share one LDS copy of the scene; 2 workgroups per CU
// render_kernel_xchg (variant 5): workgroup size and the per-wave debug record behind its error flag
#define RT_XCHG_BLOCK 768
#define RT_XCHG_DEBUG_WORDS 16u         // per wave, written behind the error flag when the kernel gives up
