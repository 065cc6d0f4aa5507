#include "rt_api.h"
#include <cstdio>
#include <vector>
extern "C" void rt_host_set_error(const char *msg) { fprintf(stderr, "err: %s\n", msg); }
int main() {
    std::vector<rt_sphere> s(64);
    uint32_t n = 0; rt_vec3 o, t;
    for (const char *p : { "/tmp/rt_san/a.scn", "/tmp/rt_san/bad1.scn", "/tmp/rt_san/bad2.scn", "/tmp/rt_san/none.scn" }) {
        int rc = rt_read_scene(p, s.data(), 64, &n, &o, &t, 1);
        printf("%s rc %d n %u\n", p, rc, n);
        rc = rt_read_scene(p, s.data(), 3, &n, &o, &t, 0);
        printf("  small cap rc %d n %u\n", rc, n);
    }
    std::vector<uint32_t> seeds(100000);
    rt_default_seeds(seeds.data(), seeds.size());
    rt_default_seeds(seeds.data(), 10);
    rt_camera c{}; c.orig = {20, 100, 120}; c.target = {0, 25, 0};
    rt_compute_camera(&c, 800, 600);
    rt_sphere d[6]; printf("demo %d %d\n", rt_demo_scene(d, 6), rt_demo_scene(d, 2));
    printf("seed0 %u cam %g\n", seeds[0], c.x.x);
}
