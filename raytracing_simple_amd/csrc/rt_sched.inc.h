// rt_sched.inc.h -- stage-scheduled form of the path-trace kernel.  Included after
// rt_trace.inc.h inside the same instance (same RT_FAST / RT_NS / option macros); needs
// RT_SCHED_KERNEL_NAME.
//
// Same arithmetic, same per-pixel operation order, different control structure.  The megakernel
// (rt_trace.inc.h) runs every section of the bounce loop in every trip for whichever lanes need
// it; the section census (tools/stamp_profile.py) shows that on the Demo scene each section is
// entered in 60-90 % of the trips with only 18-55 % of the lanes.  Here every lane carries an
// explicit stage, and in each trip the WAVEFRONT runs exactly one stage -- the one most lanes
// are waiting for (wave ballot + popcount) -- for those lanes only.  Lanes in other stages keep
// their state in registers and wait: an in-register wavefront ray queue.  Each pixel still
// consumes its own RNG stream and performs its float operations in the reference order, so the
// output stays bit-identical; only the interleaving between pixels changes.
//
//   NEW     fold the finished sample into the running average (.cl:580-589), next camera ray
//   TRACE   closest-hit sweep, hit point / normal, emission test, material dispatch
//   DIFF    next-event estimation: sample light lj (.cl:249-303); after the last light also the
//           cosine-weighted bounce (.cl:383-411) -- its draws follow the light draws, as in the
//           reference, and it does not depend on the pending shadow result
//   SHADOW  any-hit sweep of the pending shadow ray, light contribution
//   SPEC    mirror / glass bounce (.cl:413-489)
#ifndef RT_SCHED_KERNEL_NAME
#error "define RT_SCHED_KERNEL_NAME"
#endif

namespace rt {
namespace RT_NS {

enum : int { ST_NEW = 0, ST_TRACE = 1, ST_DIFF = 2, ST_SHADOW = 3, ST_SPEC = 4, ST_DONE = 5 };

extern "C" __global__ void __launch_bounds__(kBlockThreads, RT_OPT_MINWAVES)
    RT_SCHED_KERNEL_NAME(const LaunchParams P) {
    extern __shared__ float4 lds[];
    const uint32_t n = P.scene.n_spheres;
    const uint32_t n_lights = P.scene.n_lights;
    float4 *s_geom = lds;
    float4 *s_lightA = s_geom + n;
    float4 *s_lightB = s_lightA + n_lights;
    float4 *s_emis = s_lightB + n_lights;
    float4 *s_colr = s_emis + n;
    float *s_k2 = reinterpret_cast<float *>(P.mat_in_lds ? s_colr + n : s_emis);
    const bool k2_in_lds = P.n_samples <= kMaxK2Table;

    const int tid = threadIdx.x;
    __shared__ unsigned long long s_stat[5];
    if (tid < 5) s_stat[tid] = 0;
    for (uint32_t i = tid; i < n; i += kBlockThreads) s_geom[i] = P.scene.geom[i];
    for (uint32_t i = tid; i < n_lights; i += kBlockThreads) {
        s_lightA[i] = P.scene.lightA[i];
        s_lightB[i] = P.scene.lightB[i];
    }
    if (P.mat_in_lds) {
        for (uint32_t i = tid; i < n; i += kBlockThreads) {
            s_emis[i] = P.scene.emis[i];
            s_colr[i] = P.scene.colr[i];
        }
    }
    if (k2_in_lds)
        for (int i = tid; i < P.n_samples; i += kBlockThreads)
            s_k2[i] = rt_rcp((float)(P.first_sample + i) + 1.f);
#if RT_OPT_STAMPS
    __shared__ unsigned long long s_census[12];
    if (tid < 12) s_census[tid] = 0;
#endif
    __syncthreads();
    const float4 *m_emis = P.mat_in_lds ? s_emis : P.scene.emis;
    const float4 *m_colr = P.mat_in_lds ? s_colr : P.scene.colr;

    const int wave = tid >> 6, lane = tid & 63;
    const int x = blockIdx.x * kTileW + wave * 8 + (lane & 7);
    const int lrow = blockIdx.y * kTileH + (lane >> 3);
    const int tile = lrow / P.tile_rows;
    const int y = (tile * P.nranks + P.rank) * P.tile_rows + (lrow - tile * P.tile_rows);
    const bool valid = (x < P.w) && (lrow < P.local_rows) && (y < P.h);
    const size_t gid = (size_t)y * (size_t)P.w + (size_t)x;
    const size_t ci = (size_t)(P.h - y - 1) * (size_t)P.w + (size_t)x;

    uint32_t s0 = 0, s1 = 0;
    V3 acc = mk(0.f, 0.f, 0.f);
    int s = P.first_sample;
    const int s_end = P.first_sample + P.n_samples;
    if (valid) {
        s0 = P.seeds_in[2 * gid];
        s1 = P.seeds_in[2 * gid + 1];
        if (P.first_sample > 0) acc = mk(P.colors[3 * ci], P.colors[3 * ci + 1], P.colors[3 * ci + 2]);
    }

    const float inv_w = P.inv_w;
    const float inv_h = P.inv_h;
    const V3 cam_o = mk(P.cam.orig.x, P.cam.orig.y, P.cam.orig.z);
    const V3 cam_d = mk(P.cam.dir.x, P.cam.dir.y, P.cam.dir.z);
    const V3 cam_x = mk(P.cam.x.x, P.cam.x.y, P.cam.x.z);
    const V3 cam_y = mk(P.cam.y.x, P.cam.y.y, P.cam.y.z);

    uint32_t c_closest = 0, c_shadow = 0, c_draws = 0;
    unsigned long long c_tests = 0;
    unsigned long long st_roots_c = 0, st_roots_s = 0;

    // ---- per-lane path record --------------------------------------------------------------
    int stage = (valid && P.n_samples > 0) ? ST_NEW : ST_DONE;
    bool has_sample = false;          // a finished path waits to be folded into acc
    V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f);      // ray; o holds the hit point after TRACE
    V3 thr = mk(1.f, 1.f, 1.f), rad = mk(0.f, 0.f, 0.f);
    int depth = 0;
    bool after_specular = true;
    V3 nrm = mk(0.f, 0.f, 0.f);       // unoriented normal at the hit
    uint32_t id = 0;                  // sphere that was hit
    uint32_t lj = 0;                  // next light to sample
    V3 ld = mk(0.f, 0.f, 0.f);        // direct light gathered so far at this hit
    V3 sd = mk(0.f, 0.f, 0.f);        // pending shadow ray direction
    float s_len = 0.f, s_q = 0.f;     // its length and e-independent numerator (4 pi r^2 wi wo)

    for (;;) {
        // ---- pick the stage most lanes wait for (ties: the later stage) ----
        const unsigned long long b_new = __builtin_amdgcn_ballot_w64(stage == ST_NEW);
        const unsigned long long b_trc = __builtin_amdgcn_ballot_w64(stage == ST_TRACE);
        const unsigned long long b_dif = __builtin_amdgcn_ballot_w64(stage == ST_DIFF);
        const unsigned long long b_shd = __builtin_amdgcn_ballot_w64(stage == ST_SHADOW);
        const unsigned long long b_spc = __builtin_amdgcn_ballot_w64(stage == ST_SPEC);
        if ((b_new | b_trc | b_dif | b_shd | b_spc) == 0ull) break;
        int pick = ST_NEW, best = __popcll(b_new);
        { const int c = __popcll(b_trc); if (c >= best) { best = c; pick = ST_TRACE; } }
        { const int c = __popcll(b_dif); if (c >= best) { best = c; pick = ST_DIFF; } }
        { const int c = __popcll(b_shd); if (c >= best) { best = c; pick = ST_SHADOW; } }
        { const int c = __popcll(b_spc); if (c >= best) { best = c; pick = ST_SPEC; } }
        RT_STAMP(8);

        if (pick == ST_NEW) {
            if (stage == ST_NEW) {
                RT_STAMP(0);
                if (has_sample) {                                           // .cl:580-589
                    if (s == 0) {
                        acc = rad;
                    } else {
                        float k1 = (float)s;
                        float k2 = k2_in_lds ? s_k2[s - P.first_sample] : rt_rcp((float)s + 1.f);
                        acc = mk((acc.x * k1 + rad.x) * k2, (acc.y * k1 + rad.y) * k2,
                                 (acc.z * k1 + rad.z) * k2);
                    }
                    s += 1;
                    has_sample = false;
                }
                if (s >= s_end) {
                    stage = ST_DONE;
                } else {                                                    // .cl:494-549
                    float j1 = next_random(s0, s1) - 0.5f;
                    float j2 = next_random(s0, s1) - 0.5f;
                    c_draws += 2;
                    float kcx = ((float)x + j1) * inv_w - 0.5f;
                    float kcy = ((float)y + j2) * inv_h - 0.5f;
                    V3 rd = mk(cam_x.x * kcx + cam_y.x * kcy + cam_d.x,
                               cam_x.y * kcx + cam_y.y * kcy + cam_d.y,
                               cam_x.z * kcx + cam_y.z * kcy + cam_d.z);
                    o = add(scale(rd, 0.1f), cam_o);
                    d = unit(rd);
                    thr = mk(1.f, 1.f, 1.f);
                    rad = mk(0.f, 0.f, 0.f);
                    depth = 0;
                    after_specular = true;
                    stage = ST_TRACE;
                }
            }
        } else if (pick == ST_TRACE) {
            if (stage == ST_TRACE) {
                RT_STAMP(1);
                float t = 1e20f;
                id = 0;
                st_roots_c = 0;
                sweep_closest(s_geom, n, o, d, t, id, st_roots_c);          // .cl:215-232
                RT_STAMP_ROOTS(10, st_roots_c);
                c_closest += 1;
                c_tests += n;
                if (!(t < 1e20f)) {                                         // miss
                    has_sample = true;
                    stage = ST_NEW;
                } else {
                    const float4 ge = s_geom[id];
                    const float4 em4 = m_emis[id];
                    const V3 em = mk(em4.x, em4.y, em4.z);
                    const int refl = __float_as_int(em4.w);
                    V3 hp = add(o, scale(d, t));                            // .cl:338-340
                    nrm = unit(sub(hp, mk(ge.x, ge.y, ge.z)));              // .cl:345-347
                    o = hp;
                    if (!((em.x == 0.f) && (em.z == 0.f))) {                // .cl:358-368
                        if (after_specular) {
                            float dp = dot(nrm, d);
                            rad = add(rad, mul(thr, scale(em, fabsf(dp))));
                        }
                        has_sample = true;
                        stage = ST_NEW;
                    } else if (refl == RT_DIFF) {                           // .cl:370-373
                        const float4 co4 = m_colr[id];
                        after_specular = false;
                        thr = mul(thr, mk(co4.x, co4.y, co4.z));
                        ld = mk(0.f, 0.f, 0.f);
                        lj = 0;
                        stage = ST_DIFF;
                    } else {
                        stage = ST_SPEC;
                    }
                }
            }
        } else if (pick == ST_DIFF) {
            if (stage == ST_DIFF) {
                RT_STAMP(3);
                const V3 hp = o;
                const float dp = dot(nrm, d);
                const V3 nl = scale(nrm, -1.f * cl_sign(dp));               // .cl:354-355
                // ---- lights lj.. until one needs a shadow test, .cl:258-302 ----
                while (lj < n_lights) {
                    const float4 la = s_lightA[lj];
                    const float4 lb = s_lightB[lj];
                    lj += 1;
                    float u1 = next_random(s0, s1);
                    float u2 = next_random(s0, s1);
                    c_draws += 2;
                    float zc = 1.f - 2.f * u1;
                    float ring = rt_sqrt(fmaxf(0.f, 1.f - zc * zc));
                    float sphi, cphi;
#if RT_FAST
                    fm_sincos_turns(u2, sphi, cphi);
#else
                    dm_sincosf_pos((2.f * RT_PI) * u2, sphi, cphi);
#endif
                    V3 us = mk(ring * cphi, ring * sphi, zc);
                    V3 on_light = add(scale(us, la.w), mk(la.x, la.y, la.z));
                    V3 dir = sub(on_light, hp);
                    float len = rt_sqrt(dot(dir, dir));
                    dir = scale(dir, rt_rcp(len));
                    float wo = dot(dir, us);
                    if (wo > 0.f) continue;
                    wo = -wo;
                    float wi = dot(dir, nl);
                    if (wi > 0.f) {
                        sd = dir;
                        s_len = len;
                        s_q = lb.w * wi * wo;                               // numerator of .cl:297
                        stage = ST_SHADOW;
                        break;
                    }
                }
                if (lj >= n_lights) {
                    // every light has been sampled (one shadow test may still be pending):
                    // the bounce's two draws come next in the stream, .cl:383-411
                    float u = next_random(s0, s1);
                    float r2 = next_random(s0, s1);
                    c_draws += 2;
                    float r2s = rt_sqrt(r2);
                    V3 w = nl;
                    V3 a = (fabsf(w.x) > .1f) ? mk(0.f, 1.f, 0.f) : mk(1.f, 0.f, 0.f);
                    V3 uu = unit(cross(a, w));
                    V3 vv = cross(w, uu);
                    float s1v, c1v;
#if RT_FAST
                    fm_sincos_turns(u, s1v, c1v);
#else
                    dm_sincosf_pos((2.f * RT_PI) * u, s1v, c1v);
#endif
                    V3 nd = add(scale(uu, c1v * r2s), scale(vv, s1v * r2s));
                    nd = add(nd, scale(w, rt_sqrt(1 - r2)));
                    d = nd;
                    if (stage != ST_SHADOW) {                               // nothing pending
                        rad = add(rad, mul(thr, ld));                       // .cl:377-378
                        depth += 1;
                        has_sample = depth >= kMaxDepth;
                        stage = has_sample ? ST_NEW : ST_TRACE;
                    }
                }
            }
        } else if (pick == ST_SHADOW) {
            if (stage == ST_SHADOW) {
                RT_STAMP(4);
                const float max_t = s_len - RT_EPS;
                c_shadow += 1;
                st_roots_s = 0;
                const uint32_t first = sweep_any(s_geom, n, o, sd, max_t, st_roots_s);   // .cl:234-247
                RT_STAMP_ROOTS(11, st_roots_s);
                const bool blocked = first < n;
                c_tests += blocked ? first + 1 : n;
                if (!blocked) {
                    const float4 lb = s_lightB[lj - 1];
                    float k = rt_div(s_q, s_len * s_len);                   // .cl:297
                    ld = add(ld, scale(mk(lb.x, lb.y, lb.z), k));
                }
                if (lj < n_lights) {
                    stage = ST_DIFF;                                        // more lights to sample
                } else {                                                    // bounce already in d
                    rad = add(rad, mul(thr, ld));
                    depth += 1;
                    has_sample = depth >= kMaxDepth;
                    stage = has_sample ? ST_NEW : ST_TRACE;
                }
            }
        } else {
            if (stage == ST_SPEC) {
                RT_STAMP(7);
                const float4 em4 = m_emis[id];
                const float4 co4 = m_colr[id];
                const V3 col = mk(co4.x, co4.y, co4.z);
                const int refl = __float_as_int(em4.w);
                const float dp = dot(nrm, d);
                const V3 nl = scale(nrm, -1.f * cl_sign(dp));
                V3 rfl = sub(d, scale(nrm, 2.f * dot(nrm, d)));             // .cl:416-419
                after_specular = true;
                if (refl == RT_SPEC) {                                      // .cl:413-424
                    thr = mul(thr, col);
                    d = rfl;
                } else {                                                    // .cl:425-489
                    bool into = dot(nrm, nl) > 0.f;
                    const float nc = 1.f, nt = 1.52f;
                    float nnt = into ? nc / nt : nt / nc;
                    float ddn = dot(d, nl);
                    float cos2t = 1.f - nnt * nnt * (1.f - ddn * ddn);
                    if (cos2t < 0.f) {
                        thr = mul(thr, col);
                        d = rfl;
                    } else {
                        float kk = (into ? 1.f : -1.f) * (ddn * nnt + rt_sqrt(cos2t));
                        V3 td = unit(sub(scale(d, nnt), scale(nrm, kk)));
                        const float fa = nt - nc, fb = nt + nc;
                        const float R0 = fa * fa / (fb * fb);
                        float c = 1 - (into ? -ddn : dot(td, nrm));
                        float Re = R0 + (1 - R0) * c * c * c * c * c;
                        float Tr = 1.f - Re;
                        float Pr = .25f + .5f * Re;
                        float RP = rt_div(Re, Pr);
                        float TP = rt_div(Tr, 1.f - Pr);
                        float pickr = next_random(s0, s1);
                        c_draws += 1;
                        if (pickr < Pr) {
                            thr = mul(scale(thr, RP), col);
                            d = rfl;
                        } else {
                            thr = mul(scale(thr, TP), col);
                            d = td;
                        }
                    }
                }
                depth += 1;
                has_sample = depth >= kMaxDepth;
                stage = has_sample ? ST_NEW : ST_TRACE;
            }
        }
    }

    if (valid && P.n_samples > 0) {
        P.colors[3 * ci] = acc.x;
        P.colors[3 * ci + 1] = acc.y;
        P.colors[3 * ci + 2] = acc.z;
        if (!(P.skip_pixels & 1))
            P.pixels[(size_t)lrow * (size_t)P.w + (size_t)x] =
                (uint32_t)(to_int(acc.x) | (to_int(acc.y) << 8) | (to_int(acc.z) << 16));
        P.seeds[2 * gid] = s0;
        P.seeds[2 * gid + 1] = s1;
    }

    uint32_t n_done = valid ? (uint32_t)P.n_samples : 0u;
    uint32_t t_samples = wave_sum(n_done);
    uint32_t t_closest = wave_sum(c_closest);
    uint32_t t_shadow = wave_sum(c_shadow);
    uint32_t t_draws = wave_sum(c_draws);
    unsigned long long tests64 = c_tests;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tests64 += __shfl_xor(tests64, off, 64);
    // one LDS add per wavefront, then one global add per workgroup into one of kStatReplicas
    // separate lines (162 000 same-address global atomics cost 1.8 ms per launch: one word takes
    // about 88 atomics per microsecond)
    if (lane == 0) {
        atomicAdd(&s_stat[0], (unsigned long long)t_samples);
        atomicAdd(&s_stat[1], (unsigned long long)t_closest);
        atomicAdd(&s_stat[2], (unsigned long long)t_shadow);
        atomicAdd(&s_stat[3], tests64);
        atomicAdd(&s_stat[4], (unsigned long long)t_draws);
    }
    __syncthreads();
    if (tid < 5) {
        const unsigned block_linear = blockIdx.x + blockIdx.y * gridDim.x;
        atomicAdd(&P.stats[(block_linear % (unsigned)kStatReplicas) * 8u + (unsigned)tid], s_stat[tid]);
    }
#if RT_OPT_STAMPS
    __syncthreads();
    if (tid < 12) atomicAdd(&P.counters[8 + tid], s_census[tid]);
#endif
}

}  // namespace RT_NS
}  // namespace rt
