#include "rt_api.h"
#include <cstdio>
#include <string>
#include <vector>
extern "C" void rt_host_set_error(const char *) {}
int main() {
    std::vector<rt_sphere> s(40000);
    int ok = 0, err = 0;
    for (int i = 0; i < 1000; ++i) {
        uint32_t n = 0; rt_vec3 o, t;
        std::string p = "/tmp/rt_san/fz/f" + std::to_string(i) + ".scn";
        for (int dbl = 0; dbl < 2; ++dbl) {
            int rc = rt_read_scene(p.c_str(), s.data(), dbl ? 40000 : 7, &n, &o, &t, dbl);
            (rc == 0 ? ok : err)++;
        }
    }
    printf("reader fuzz: %d ok, %d rejected, no crash\n", ok, err);
}
