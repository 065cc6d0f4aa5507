#include "rt_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(void) {
    int w = 96, h = 64, spp = 6;
    orc_sphere sph[6] = {
        { 1000.f, { 0.f, -1000.f, 0.f }, { 0.f, 0.f, 0.f }, { 0.75f, 0.75f, 0.75f }, 0 },
        { 12.f, { 40.f, 20.f, 0.f }, { 0.f, 0.f, 0.f }, { 0.9f, 0.f, 0.f }, 2 },
        { 11.f, { -35.f, 20.f, 0.f }, { 0.f, 0.f, 0.f }, { 0.f, 0.9f, 0.f }, 2 },
        { 10.f, { 0.f, 25.f, -10.f }, { 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.9f }, 1 },
        { 9.f, { 20.f, 10.f, -5.f }, { 0.f, 0.f, 0.f }, { 0.9f, 0.f, 0.9f }, 0 },
        { 7.f, { 0.f, 60.f, 0.f }, { 12.f, 12.f, 12.f }, { 0.f, 0.f, 0.f }, 0 },
    };
    orc_camera cam; memset(&cam, 0, sizeof cam);
    cam.orig.x = 20.f; cam.orig.y = 100.f; cam.orig.z = 120.f; cam.target.y = 25.f;
    orc_camera_basis(&cam, w, h);
    uint32_t *seeds = malloc(sizeof(uint32_t) * 2 * w * h), *pix = calloc(w * h, 4);
    orc_vec *col = calloc(w * h, sizeof(orc_vec));
    orc_seeds_init(seeds, w, h);
    orc_stats st; memset(&st, 0, sizeof st);
    orc_render(col, seeds, sph, 6, &cam, w, h, 0, spp, pix, 4, &st);
    printf("fnv pixels %016llx\n", (unsigned long long)orc_fnv1a64(pix, 4u * w * h));
    free(seeds); free(pix); free(col);
    return 0;
}
