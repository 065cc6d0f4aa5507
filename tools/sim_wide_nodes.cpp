// How many steps a ray's walk takes through the hierarchy of a large scene as sibling PAIRS (what rt_walk.inc.h walks) and as
// 4-wide nodes (two levels of the same tree collapsed): a host-side model, to decide whether 4-wide nodes are worth building.
// Same tree shape as rt_bvh.hip (leaves of 8, leaf ranges split in the middle, spheres sorted along the longest axis of the
// box of their centres); rays: the scene's camera rays in 8x8 tiles, one diffuse bounce ray and one shadow ray from each hit.
// A wavefront is 64 consecutive rays; it is busy for max-over-lanes steps.
//   g++ -O2 -o /tmp/sim_wide_nodes tools/sim_wide_nodes.cpp && python tools/sim_wide_nodes.py | /tmp/sim_wide_nodes
// stdin: n, then n x (rad, px, py, pz), then camera orig(3) dir(3) x(3) y(3), w, h
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

struct S { float r, x, y, z; };
struct Box { float lo[3], hi[3]; };
static std::vector<S> sph;              // tree spheres, in leaf order after the build
static std::vector<Box> leaf_box;
static int n_leaves;

static Box box_of(int a, int b) {       // leaves [a, b)
    Box bx{{1e30f, 1e30f, 1e30f}, {-1e30f, -1e30f, -1e30f}};
    for (int l = a; l < b; ++l)
        for (int k = 0; k < 3; ++k) { bx.lo[k] = std::min(bx.lo[k], leaf_box[l].lo[k]); bx.hi[k] = std::max(bx.hi[k], leaf_box[l].hi[k]); }
    return bx;
}
static void order(int first, int last, int la, int lb) {        // spheres [first, last) go to leaves [la, lb)
    if (lb - la <= 1) return;
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = first; i < last; ++i) {
        const float c[3] = {sph[i].x, sph[i].y, sph[i].z};
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], c[k]); hi[k] = std::max(hi[k], c[k]); }
    }
    int ax = 0;
    for (int k = 1; k < 3; ++k) if (hi[k] - lo[k] > hi[ax] - lo[ax]) ax = k;
    std::sort(sph.begin() + first, sph.begin() + last, [ax](const S &p, const S &q) { return (&p.x)[ax] < (&q.x)[ax]; });
    const int mid = (la + lb) / 2, cut = std::min(first + (mid - la) * 8, last);
    order(first, cut, la, mid);
    order(cut, last, mid, lb);
}
static bool hit_box(const Box &b, const float o[3], const float inv[3], float far, float &tn) {
    float t0 = 0.f, t1 = far;
    for (int k = 0; k < 3; ++k) {
        float a = (b.lo[k] - o[k]) * inv[k], c = (b.hi[k] - o[k]) * inv[k];
        if (a > c) std::swap(a, c);
        t0 = std::max(t0, a); t1 = std::min(t1, c);
    }
    tn = t0;
    return t0 <= t1;
}
static float hit_sphere(const S &s, const float o[3], const float d[3]) {
    const float op[3] = {s.x - o[0], s.y - o[1], s.z - o[2]};
    const float b = op[0] * d[0] + op[1] * d[1] + op[2] * d[2];
    float det = b * b - (op[0] * op[0] + op[1] * op[1] + op[2] * op[2]) + s.r * s.r;
    if (det < 0) return 0;
    det = std::sqrt(det);
    const float t1 = b - det, t2 = b + det;
    return t1 > 0.01f ? t1 : (t2 > 0.01f ? t2 : 0.f);
}
struct Counts { long steps = 0, leaves = 0, boxes = 0; };
// leaf test; returns the new bound
static float do_leaf(int l, const float o[3], const float d[3], float far, bool any, bool &done) {
    for (int i = l * 8; i < std::min((int)sph.size(), l * 8 + 8); ++i) {
        const float t = hit_sphere(sph[i], o, d);
        if (t > 0 && t < far) { far = t; if (any) done = true; }
    }
    return far;
}
// width 2: a step = the two children of an inner node; width 4: its (up to) four grandchildren
static float walk(int width, const float o[3], const float d[3], float far, bool any, Counts &c) {
    float inv[3];
    for (int k = 0; k < 3; ++k) inv[k] = 1.f / d[k];
    struct N { int a, b; };
    std::vector<N> st;
    if (n_leaves == 1) { bool dn = false; c.leaves++; return do_leaf(0, o, d, far, any, dn); }
    st.push_back({0, n_leaves});
    bool done = false;
    while (!st.empty() && !done) {
        N nd = st.back(); st.pop_back();
        if (nd.b - nd.a == 1) { c.leaves++; far = do_leaf(nd.a, o, d, far, any, done); continue; }
        c.steps++;
        N kids[4]; int nk = 0;
        const int mid = (nd.a + nd.b) / 2;
        const N two[2] = {{nd.a, mid}, {mid, nd.b}};
        for (const N &t : two) {
            if (width == 4 && t.b - t.a > 1) { const int m2 = (t.a + t.b) / 2; kids[nk++] = {t.a, m2}; kids[nk++] = {m2, t.b}; }
            else kids[nk++] = t;
        }
        float tn[4]; bool ok[4];
        for (int k = 0; k < nk; ++k) { c.boxes++; ok[k] = hit_box(box_of(kids[k].a, kids[k].b), o, inv, far, tn[k]); }
        int idx[4] = {0, 1, 2, 3};
        std::sort(idx, idx + nk, [&](int p, int q) { return tn[p] > tn[q]; });     // farthest pushed first
        for (int k = 0; k < nk; ++k) if (ok[idx[k]]) st.push_back(kids[idx[k]]);
    }
    return far;
}

int main() {
    int n; if (scanf("%d", &n) != 1) return 1;
    std::vector<S> all(n);
    for (auto &s : all) if (scanf("%f %f %f %f", &s.r, &s.x, &s.y, &s.z) != 4) return 1;
    float cam[12]; int w, h;
    for (float &v : cam) if (scanf("%f", &v) != 1) return 1;
    if (scanf("%d %d", &w, &h) != 2) return 1;
    std::vector<S> always;
    std::vector<float> rr;
    for (auto &s : all) rr.push_back(std::fabs(s.r));
    std::nth_element(rr.begin(), rr.begin() + n / 2, rr.end());
    const float r_cut = 8.f * rr[n / 2];
    for (auto &s : all) (std::fabs(s.r) <= r_cut ? sph : always).push_back(s);
    n_leaves = ((int)sph.size() + 7) / 8;
    order(0, (int)sph.size(), 0, n_leaves);
    leaf_box.resize(n_leaves);
    for (int l = 0; l < n_leaves; ++l) {
        Box b{{1e30f, 1e30f, 1e30f}, {-1e30f, -1e30f, -1e30f}};
        for (int i = l * 8; i < std::min((int)sph.size(), l * 8 + 8); ++i) {
            const float c[3] = {sph[i].x, sph[i].y, sph[i].z};
            for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], c[k] - sph[i].r); b.hi[k] = std::max(b.hi[k], c[k] + sph[i].r); }
        }
        leaf_box[l] = b;
    }
    // box_of() is O(leaves) per call: cache the boxes of all nodes by (a, b) -- small scenes only; fine for a model
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    struct Ray { float o[3], d[3], far; bool any; };
    std::vector<Ray> prim, bounce, shadow;
    for (int ty = 0; ty < h; ty += 8 * 6)
        for (int tx = 0; tx < w; tx += 8 * 6)
            for (int y = ty; y < ty + 8; ++y)
                for (int x = tx; x < tx + 8; ++x) {
                    const float kx = (x + 0.5f) / w - 0.5f, ky = (y + 0.5f) / h - 0.5f;
                    Ray r; float len = 0;
                    for (int k = 0; k < 3; ++k) { r.d[k] = cam[6 + k] * kx + cam[9 + k] * ky + cam[3 + k]; len += r.d[k] * r.d[k]; }
                    for (int k = 0; k < 3; ++k) { r.o[k] = cam[k] + 0.1f * r.d[k]; r.d[k] /= std::sqrt(len); }
                    r.far = 1e20f; r.any = false;
                    prim.push_back(r);
                }
    auto trace_all = [&](const std::vector<Ray> &rays, const char *name, std::vector<float> *t_out) {
        for (int width : {2, 4}) {
            Counts tot; long wave_steps = 0, wave_leaves = 0, waves = 0;
            for (size_t base = 0; base < rays.size(); base += 64) {
                long ms = 0, ml = 0;
                for (size_t i = base; i < std::min(rays.size(), base + 64); ++i) {
                    Counts c;
                    float far = rays[i].far;
                    for (auto &s : always) { const float t = hit_sphere(s, rays[i].o, rays[i].d); if (t > 0 && t < far) far = t; }
                    const float t = walk(width, rays[i].o, rays[i].d, far, rays[i].any, c);
                    if (t_out && width == 2) (*t_out)[i] = t;
                    tot.steps += c.steps; tot.leaves += c.leaves; tot.boxes += c.boxes;
                    ms = std::max(ms, c.steps); ml = std::max(ml, c.leaves);
                }
                wave_steps += ms; wave_leaves += ml; waves++;
            }
            printf("%-8s width %d: per ray %.2f steps, %.2f box tests, %.2f leaves; per wavefront (max over 64 lanes) %.1f steps, %.1f leaves\n", name, width,
                   (double)tot.steps / rays.size(), (double)tot.boxes / rays.size(), (double)tot.leaves / rays.size(), (double)wave_steps / waves,
                   (double)wave_leaves / waves);
        }
    };
    std::vector<float> t(prim.size());
    trace_all(prim, "primary", &t);
    const float light[3] = {0.f, 60.f, 0.f};
    for (size_t i = 0; i < prim.size(); ++i) {
        if (!(t[i] < 1e19f)) continue;
        Ray b; float hp[3];
        for (int k = 0; k < 3; ++k) hp[k] = prim[i].o[k] + prim[i].d[k] * t[i];
        // cosine-weighted around the surface normal at the hit (the sphere whose surface the point lies on; else the ground's)
        float nrm[3] = {0.f, 1.f, 0.f};
        for (const S &q : sph) {
            const float v[3] = {hp[0] - q.x, hp[1] - q.y, hp[2] - q.z};
            const float dist = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            if (std::fabs(dist - q.r) < 1e-3f * q.r + 1e-3f) { for (int k = 0; k < 3; ++k) nrm[k] = v[k] / dist; break; }
        }
        float uu[3] = {nrm[2], 0.f, -nrm[0]};                        // (0,1,0) x n, or (1,0,0) x n for a vertical normal
        if (std::fabs(nrm[0]) < 0.1f && std::fabs(nrm[2]) < 0.1f) { uu[0] = 0.f; uu[1] = -nrm[2]; uu[2] = nrm[1]; }
        const float ul = std::sqrt(uu[0] * uu[0] + uu[1] * uu[1] + uu[2] * uu[2]);
        for (float &v : uu) v /= ul;
        const float vv[3] = {nrm[1] * uu[2] - nrm[2] * uu[1], nrm[2] * uu[0] - nrm[0] * uu[2], nrm[0] * uu[1] - nrm[1] * uu[0]};
        const float r1 = 6.2831853f * U(rng), r2 = U(rng), r2s = std::sqrt(r2), cz = std::sqrt(1 - r2);
        for (int k = 0; k < 3; ++k) b.d[k] = uu[k] * std::cos(r1) * r2s + vv[k] * std::sin(r1) * r2s + nrm[k] * cz;
        for (int k = 0; k < 3; ++k) b.o[k] = hp[k] + 0.02f * nrm[k];
        b.far = 1e20f; b.any = false;
        bounce.push_back(b);
        Ray s; float len = 0;
        for (int k = 0; k < 3; ++k) { s.d[k] = light[k] - hp[k]; len += s.d[k] * s.d[k]; }
        len = std::sqrt(len);
        for (int k = 0; k < 3; ++k) { s.d[k] /= len; s.o[k] = hp[k] + 0.02f * s.d[k]; }
        s.far = len - 7.f; s.any = true;
        shadow.push_back(s);
    }
    trace_all(bounce, "bounce", nullptr);
    trace_all(shadow, "shadow", nullptr);
    return 0;
}
