/* A C99 client of include/gsx.h, as a maintainer of a C / C++ viewer would write it: links against libgsx.so and calls what
 * needs no GPU -- version, parameter defaults with the struct sizes THIS translation unit compiled, host-side size arithmetic,
 * argument errors.  Built and run by tests/test_cabi.py (gcc; no HIP headers needed: the ABI is plain C). */
#include <stdio.h>
#include <string.h>

#include "gsx.h"

int main(void) {
    unsigned char raw[sizeof(GsxParams) + 64];
    GsxParams *p = (GsxParams *)raw;
    GsxFrameStats st;
    GsxCamera cam;
    size_t k;
    int rc;
    if (gsx_version() != GSX_VERSION) { printf("version %d != header %d\n", gsx_version(), GSX_VERSION); return 1; }
    memset(raw, 0xAB, sizeof raw);
    gsx_default_params(p);                                   /* the macro: passes sizeof(GsxParams) */
    for (k = sizeof(GsxParams); k < sizeof raw; ++k)
        if (raw[k] != 0xAB) { printf("default_params wrote behind the struct at byte %u\n", (unsigned)k); return 2; }
    if (p->struct_size != (int)sizeof(GsxParams) || p->stats_size != (int)sizeof(GsxFrameStats) || p->tile_x1 != -1 ||
        p->semantics != GSX_SEM_REF_CPU || p->original_index != NULL || p->block_bounds != NULL || p->row_of_index != NULL) {
        printf("unexpected defaults\n");
        return 3;
    }
    if (sizeof(GsxFrameStats) != 72 || GSX_FRAME_STATS_BYTES_ABI300 != 64) { printf("stats struct size\n"); return 4; }
    if (gsx_workspace_bytes(1000000, 1920, 1080, 16, 5000000) == 0 || gsx_workspace_bytes(-1, 1920, 1080, 16, 1) != 0 ||
        gsx_hints_bytes(1920, 1080, 16) == 0) { printf("size arithmetic\n"); return 5; }
    memset(&cam, 0, sizeof cam);
    cam.width = 64; cam.height = 64;
    memset(&st, 0, sizeof st);
    /* no workspace: refused before anything touches a device, with a message */
    rc = gsx_render_forward(&cam, NULL, NULL, NULL, NULL, NULL, 0, 16, (float *)256, p, &st, NULL, 0, NULL);
    if (rc != GSX_ERR_INVALID_ARGUMENT || strstr(gsx_last_error(), "workspace") == NULL) { printf("error path: %d %s\n", rc, gsx_last_error()); return 6; }
    p->original_index = (const int32_t *)256;                /* reordered rows without the inverse permutation: refused */
    rc = gsx_render_forward(&cam, (const float *)256, (const float *)256, (const float *)256, (const float *)256, (const float *)256,
                            1, 16, (float *)256, p, &st, (void *)256, 1u << 20, NULL);
    if (rc != GSX_ERR_INVALID_ARGUMENT || strstr(gsx_last_error(), "row_of_index") == NULL) { printf("row_of_index: %d %s\n", rc, gsx_last_error()); return 7; }
    printf("gsx C ABI %d: GsxParams %u bytes, GsxFrameStats %u bytes, GsxCamera %u bytes\n", gsx_version(), (unsigned)sizeof(GsxParams),
           (unsigned)sizeof(GsxFrameStats), (unsigned)sizeof(GsxCamera));
    return 0;
}
