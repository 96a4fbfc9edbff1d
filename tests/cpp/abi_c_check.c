/* The boundary must be usable from plain C: this file includes include/rt06.h as C11 and touches every struct. */
#include <stdio.h>
#include "rt06.h"

int main(void) {
    rt_bvh_node n; rt_prim p; rt_material m; rt_world_flat w; rt_camera c; rt_render_config cfg; rt_quad q; (void)q;
    (void)n; (void)p; (void)m; (void)w; (void)c; (void)cfg;
    if (sizeof(rt_bvh_node) != 32 || sizeof(rt_prim) != 32 || sizeof(rt_material) != 32 || sizeof(rt_camera) != 76 || sizeof(rt_quad) != 80 || sizeof(rt_world_flat) != 128 || sizeof(rt_perlin) != 6144) return 1;
    rt_scene* s = NULL;
    if (rt_scene_three_spheres(&s) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return 2; }
    if (rt_scene_get_flat(s, &w) != RT_OK || w.kind != RT_WORLD_LIST || w.n_prims != 5) return 3;
    rt_scene_destroy(s);
    if (rt_scene_get_flat(NULL, &w) == RT_OK || rt_last_error()[0] == 0) return 4; /* never silent */
    printf("C ABI ok: %s\n", rt_version());
    return 0;
}
