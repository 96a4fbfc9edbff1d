// rt_internal.hpp — declarations shared by the host and device translation units of librt06.so.
#pragma once
#include <stdint.h>

#define RT_MAX_STACK 32  // _PRIO_QUEUE_ELEM_COUNT, rt_engine/geometry/BVH.cu:17
#define RT_TILE 8        // 8x8-pixel tiles: the reference's thread-block shape, Renderer.cu:130

int rt_fail(int code, const char* fmt, ...);
