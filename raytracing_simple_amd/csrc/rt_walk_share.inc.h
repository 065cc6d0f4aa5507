// rt_walk_share.inc.h -- the hierarchy walk with the wavefront's lanes sharing the rays' work (RT_OPT_WALK 3), included by
// rt_walk.inc.h in place of its kernel; the ray set-up and the box test above it there are shared.
//
// rt_walk.inc.h's kernel walks one ray per lane and a trip's walk phase lasts as long as its longest walk: on C3 41 % of the
// pair steps and 49 % of the leaf steps run with 8 lanes or fewer (profiles/r04b_walk_census_lanes_per_step.jsonl).  Here a
// lane whose walk has ended TAKES OVER part of a neighbour's: the oldest sibling the neighbour has kept -- the root of a whole
// subtree the ray's walk still has to visit -- together with the ray.  What a lane walks is therefore a PIECE: a ray (its own
// or another lane's), a subtree of it, and what is known about the ray so far.
//
//   * Results meet in one 64-bit word per ray in LDS (s_box[the ray's lane]), combined with ds_min_u64:
//       closest hit  { bits(t), scene index }   -- the 64-bit minimum IS the reference's rule (.cl:215-232: strictly nearer wins,
//                                                  scene order breaks ties), whichever piece finds what, in whatever order;
//       shadow ray   { lowest blocking scene index, 0 }   (.cl:234-247 returns at the first blocker in scene order).
//     A piece posts every candidate that is at least as good as what it knows, at the end of the leaf it was found in; nothing
//     is kept per lane but the distance (closest) or the index (shadow) it prunes with -- and before every round of steps a piece
//     reads the word's upper half again, so what any piece of the ray has found bounds all of them.
//   * The scene index of a slot comes from LDS (u16: a tree that fits LDS has fewer than 65536 spheres), so a candidate costs an
//     LDS read, not a trip to L2; the hit's geometry and material are read by scene index from the scene tables.
//   * Taking over: when `take_min` lanes of the wavefront have nothing to walk and some lane has kept siblings, idle lane number j
//     pairs with giver number j (ranks by mbcnt, the pairing through 64 words of LDS), reads the giver's piece -- 19 registers,
//     by ds_bpermute, no LDS storage -- and the giver's OLDEST kept sibling from the giver's stack column; the giver moves its
//     stack's base up by one.  The oldest is the largest subtree still to do.  A lane that takes nothing reads itself.
//
// Every candidate still goes through the reference's test and the winner is chosen by the reference's rule: frames, seeds and
// counters equal the plain sweep's.  A piece may test MORE candidates than the single walk would (its bound is what was known
// when it last looked) -- the counters do not count candidates (sphere_tests is what the reference's loop would have done).
constexpr uint32_t kNoIndex = 0xffffffffu;

RT_DEV float lane_read(int src, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v))); }
RT_DEV uint32_t lane_read(int src, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)v); }
RT_DEV int lane_rank(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

struct Piece {            // what a lane walks: its own ray from the root, or a subtree of another lane's ray
    V3 o, d;
    BvhRay R;
    uint32_t cur;         // a pair, kBvhLeafRef | leaf, or kWalkDone
    int sp, base;         // its kept siblings: entries base .. sp - 1 of the lane's stack column
    float far;            // closest hit: the best distance known for the ray; shadow ray: its length
    uint32_t bnd;         // shadow ray: the lowest blocking scene index known (closest hit: kNoIndex, prunes nothing)
    uint32_t own;         // the ray's lane in the workgroup << 1 | shadow
};

// The walk phase of a trip: until no lane of the wavefront has anything left to walk.  Called by all 64 lanes.
// cen (census instance only): [0]/[1] pair steps of the wavefront / lane steps, [2]/[3] leaf steps, [4] take-over phases, [5] pieces taken
RT_DEV void walk_shared(const float4 *s_pairs, const float4 *s_slots, const uint16_t *s_index, uint16_t *s_stack, unsigned long long *s_box,
                        uint32_t *s_match, uint32_t n_always, int tid, int round_len, int take_min, int take_passes, Piece &W, unsigned long long *cen) {
    constexpr int stride = 64 * RT_OPT_WG_WAVES;
    uint16_t *my_stack = s_stack + tid;
    uint32_t *match = s_match + (tid & ~63);
    for (;;) {
        bool walking = W.cur != kWalkDone;
        const unsigned long long bw = __builtin_amdgcn_ballot_w64(walking);
        if (bw == 0ull) break;
        // ---- take over: lanes with nothing to walk each take the oldest kept sibling of a lane that has one ----
        // (up to `take_passes` pairings in a row: a giver hands out one sibling per pairing, and a walk that has kept several feeds several lanes)
        for (int pass = 0; pass < take_passes; ++pass) {
            const unsigned long long bw2 = __builtin_amdgcn_ballot_w64(walking);
            const bool rich = walking & (W.sp > W.base);
            const unsigned long long br = __builtin_amdgcn_ballot_w64(rich);
            const int n_idle = 64 - __popcll(bw2);
            if (n_idle < take_min || br == 0ull) break;
            const int n_rich = __popcll(br);
            const int k = n_idle < n_rich ? n_idle : n_rich;
            const int r_rich = lane_rank(br), r_idle = lane_rank(~bw2);
            const bool give = rich & (r_rich < k), take = !walking & (r_idle < k);
            if (give) match[r_rich] = (uint32_t)tid;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t from = (uint32_t)tid;
            if (take) from = match[r_idle];
            const int src = (int)((from & 63u) << 2);
            W.o = mk(lane_read(src, W.o.x), lane_read(src, W.o.y), lane_read(src, W.o.z));
            W.d = mk(lane_read(src, W.d.x), lane_read(src, W.d.y), lane_read(src, W.d.z));
            W.R.clo = mk(lane_read(src, W.R.clo.x), lane_read(src, W.R.clo.y), lane_read(src, W.R.clo.z));
            W.R.chi = mk(lane_read(src, W.R.chi.x), lane_read(src, W.R.chi.y), lane_read(src, W.R.chi.z));
            W.R.inv = mk(lane_read(src, W.R.inv.x), lane_read(src, W.R.inv.y), lane_read(src, W.R.inv.z));
            W.R.tback = lane_read(src, W.R.tback);
            W.far = lane_read(src, W.far);
            W.bnd = lane_read(src, W.bnd);
            W.own = lane_read(src, W.own);
            const int base_from = (int)lane_read(src, (uint32_t)W.base);
            if (take) {
                W.cur = (uint32_t)s_stack[base_from * stride + (int)from];
                W.sp = 0;
                W.base = 0;
            }
            W.base += give ? 1 : 0;
            walking = walking | take;
            __builtin_amdgcn_wave_barrier();
            if (cen) {
                if ((tid & 63) == 0) cen[4] += 1ull;
                cen[5] += take ? 1ull : 0ull;
            }
        }
        if (walking) {
            const bool shadow = (W.own & 1u) != 0u;
            unsigned long long *box = s_box + (W.own >> 1);
            {
                // what the ray's pieces have found since this one last looked
                const uint32_t seen = reinterpret_cast<const uint32_t *>(box)[1];
                W.bnd = shadow ? (seen < W.bnd ? seen : W.bnd) : W.bnd;
                W.far = shadow ? W.far : fminf(W.far, __uint_as_float(seen));
            }
            for (int round = round_len; W.cur < kBvhLeafRef && round > 0; --round) {
                if (cen) {
                    const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true);
                    if ((tid & 63) == __ffsll((long long)act_) - 1) cen[0] += 1ull;
                    cen[1] += 1ull;
                }
                const float4 *pp = s_pairs + 4u * W.cur;
                const float4 A0 = pp[0], B0 = pp[1], A1 = pp[2], B1 = pp[3];
                float tn0, tn1;
                const bool out0 = bvh_misses(W.R, A0, B0, W.far, tn0), out1 = bvh_misses(W.R, A1, B1, W.far, tn1);
                // a shadow walk skips subtrees that hold only scene indices above its lowest blocker so far (a closest-hit walk's bound is ~0)
                const bool m0 = (int)out0 | (int)(__float_as_uint(B0.w) > W.bnd), m1 = (int)out1 | (int)(__float_as_uint(B1.w) > W.bnd);
                const uint32_t r0 = __float_as_uint(A0.w), r1 = __float_as_uint(A1.w);
                const bool both = !m0 & !m1, none = m0 & m1;
                const bool second_first = both ? (tn1 < tn0) : m0;
                const uint32_t near = second_first ? r1 : r0, far = second_first ? r0 : r1;
                my_stack[W.sp * stride] = (uint16_t)far;       // (dead unless `both`: the entry above the top)
                W.sp += both ? 1 : 0;
                if (none) {
                    W.sp -= 1;
                    W.cur = W.sp >= W.base ? (uint32_t)my_stack[W.sp * stride] : kWalkDone;
                    W.sp = W.sp < W.base ? W.base : W.sp;
                } else {
                    W.cur = near;
                }
            }
            if (W.cur != kWalkDone && W.cur >= kBvhLeafRef) {
                if (cen) {
                    const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true);
                    if ((tid & 63) == __ffsll((long long)act_) - 1) cen[2] += 1ull;
                    cen[3] += 1ull;
                }
                const uint32_t sl = n_always + (uint32_t)kBvhLeaf * (W.cur & (kBvhLeafRef - 1u));
                // the leaf's spheres whose scene index is wanted: a shadow ray's blockers; for a closest hit the spheres at exactly the best
                // distance -- a strictly nearer one clears the mask and enters it alone
                uint32_t want = 0u;
#pragma unroll
                for (int half = 0; half < kBvhLeaf; half += 4) {
                    HitPre p[4];
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) p[k4] = hit_pre(s_slots[sl + (uint32_t)(half + k4)], W.o, W.d);
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const int k = half + k4;
                        if (wave_any_nonneg(p[k4].det)) {
                            const HitRoots hr = hit_roots(p[k4]);
                            const bool nearer = hr.hit & (hr.t < W.far), level = hr.hit & (hr.t == W.far);
                            const bool take = nearer & !shadow;
                            want = take ? 0u : want;
                            want |= (nearer | (level & !shadow)) ? (1u << k) : 0u;
                            W.far = take ? hr.t : W.far;
                        }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(want != 0u) != 0ull) {
                    if (want != 0u) {
                        uint32_t best = kNoIndex;
                        do {
                            const uint32_t k = (uint32_t)__builtin_ctz(want);
                            want &= want - 1u;
                            const uint32_t ix = (uint32_t)s_index[sl + k];
                            best = ix < best ? ix : best;
                        } while (want != 0u);
                        const unsigned long long key = shadow ? ((unsigned long long)best << 32)
                                                              : (((unsigned long long)__float_as_uint(W.far) << 32) | (unsigned long long)best);
                        atomicMin(box, key);
                        W.bnd = shadow ? (best < W.bnd ? best : W.bnd) : W.bnd;
                    }
                }
                W.sp -= 1;
                W.cur = W.sp >= W.base ? (uint32_t)my_stack[W.sp * stride] : kWalkDone;
                W.sp = W.sp < W.base ? W.base : W.sp;
            }
        }
    }
}

extern "C" __global__ void __launch_bounds__(64 * RT_OPT_WG_WAVES, RT_OPT_MINWAVES) RT_KERNEL_NAME(const LaunchParams P) {
    constexpr int kBlockThreads = 64 * RT_OPT_WG_WAVES;
    constexpr int kTileW = 8 * RT_OPT_WG_WAVES;
    extern __shared__ float4 lds[];
    const uint32_t n = P.scene.n_spheres;
    const uint32_t n_lights = P.scene.n_lights;
    const uint32_t n_always = P.bvh.n_always, n_slots = P.bvh.n_slots;
    float4 *s_hdr = lds;
    const uint32_t n_pairs = P.bvh.n_leaves - 1u;
    const uint32_t stack_f4 = (P.bvh.stack_depth * (uint32_t)kBlockThreads * 2u + 15u) / 16u;
    const uint32_t index_f4 = (n_slots * 2u + 15u) / 16u;
    const uint32_t root_ref = n_pairs ? P.bvh.root : kBvhLeafRef;
    // staged: hdr | pairs | slots | scene index of every slot (u16) | one stack of P.bvh.stack_depth u16 per lane ([level][lane]) | lights
    float4 *s_pairs = s_hdr + 2;
    float4 *s_slots = s_pairs + 4 * n_pairs;
    uint16_t *s_index = reinterpret_cast<uint16_t *>(s_slots + n_slots);
    uint16_t *s_stack = reinterpret_cast<uint16_t *>(s_slots + n_slots + index_f4);
    float4 *s_lightA = s_slots + n_slots + index_f4 + stack_f4;     // {centre, radius}
    float4 *s_lightB = s_lightA + n_lights;                         // {emission, 4*pi*radius^2}
    float *s_k2 = reinterpret_cast<float *>(s_lightB + n_lights);
    const bool k2_in_lds = P.n_samples <= kMaxK2Table;

    const int tid = threadIdx.x;
    __shared__ unsigned long long s_stat[5];
    __shared__ unsigned s_tile_cost;
    __shared__ float4 s_cam[4];         // orig, dir | x, y | 1/w, 1/h
    __shared__ unsigned long long s_box[kBlockThreads];     // per ray (= per lane): what its pieces have found (above)
    __shared__ uint32_t s_match[kBlockThreads];             // per wavefront: the givers of a take-over phase, by rank
    if (tid < 5) s_stat[tid] = 0;
    if (tid == 5) s_tile_cost = 0u;
    if (tid == 6) {
        s_cam[0] = make_float4(P.cam.orig.x, P.cam.orig.y, P.cam.orig.z, P.cam.dir.x);
        s_cam[1] = make_float4(P.cam.dir.y, P.cam.dir.z, P.cam.x.x, P.cam.x.y);
        s_cam[2] = make_float4(P.cam.x.z, P.cam.y.x, P.cam.y.y, P.cam.y.z);
        s_cam[3] = make_float4(0.f, 0.f, P.inv_w, P.inv_h);
    }
    if (tid < 2) s_hdr[tid] = P.bvh.blob[tid];
    {
        const float4 *g_pairs = P.bvh.blob + bvh_pairs_at(n_slots);
        const float4 *g_slots = P.bvh.blob + bvh_slots_at();
        const uint32_t *g_index = reinterpret_cast<const uint32_t *>(P.bvh.blob + bvh_index_at(n_slots));
        for (uint32_t i = tid; i < 4u * n_pairs; i += kBlockThreads) s_pairs[i] = g_pairs[i];
        for (uint32_t i = tid; i < n_slots; i += kBlockThreads) {
            s_slots[i] = g_slots[i];
            s_index[i] = (uint16_t)g_index[i];
        }
    }
    for (uint32_t i = tid; i < n_lights; i += kBlockThreads) {
        s_lightA[i] = P.scene.lightA[i];
        s_lightB[i] = P.scene.lightB[i];
    }
    if (k2_in_lds)
        for (int i = tid; i < P.n_samples; i += kBlockThreads) s_k2[i] = rt_rcp((float)(P.first_sample + i) + 1.f);
    __syncthreads();

    // ---- pixel of this lane (as in rt_trace.inc.h) ------------------------------------------
    const int wave = tid >> 6, lane = tid & 63;
    const unsigned block_linear = blockIdx.x + blockIdx.y * gridDim.x;
    const unsigned tile_id = P.order ? P.order[block_linear] : block_linear;
    const int tile_by = (int)(tile_id / gridDim.x), tile_bx = (int)(tile_id - (unsigned)tile_by * gridDim.x);
    __shared__ unsigned long long s_wave_t0[RT_OPT_WG_WAVES];
    if (lane == 0) s_wave_t0[wave] = __builtin_amdgcn_s_memrealtime();
    int x = tile_bx * kTileW + wave * 8 + (lane & 7), lrow = tile_by * kTileH + (lane >> 3);
    if (P.deal) {
        const int bands = P.deal_rows >> 3, region_y = tile_by / bands, band = tile_by - region_y * bands;
        const unsigned id = P.deal[(size_t)(region_y * (int)gridDim.x + tile_bx) * (size_t)(kRegionW * P.deal_rows) + (unsigned)(band * 256 + tid)];
        x = tile_bx * kTileW + (int)(id & 31u);
        lrow = region_y * P.deal_rows + (int)(id >> 5);
    }
    const int rtile = lrow / P.tile_rows;
    const int y = (rtile * P.nranks + P.rank) * P.tile_rows + (lrow - rtile * P.tile_rows);
    const bool valid = (x < P.w) && (lrow < P.local_rows) && (y < P.h);
    const uint32_t xy = (uint32_t)x | ((uint32_t)y << 16);

    uint32_t s0 = 0, s1 = 0;
    V3 acc = mk(0.f, 0.f, 0.f);
    int s = P.first_sample;
    const int s_end = valid ? P.first_sample + P.n_samples : P.first_sample;
    if (valid) {
        const size_t gid = (size_t)y * (size_t)P.w + (size_t)x;             // .cl:560-563
        const size_t ci = (size_t)(P.h - y - 1) * (size_t)P.w + (size_t)x;  // .cl:579
        const uint2 sd = *reinterpret_cast<const uint2 *>(P.seeds_in + 2 * gid);
        s0 = sd.x;
        s1 = sd.y;
        if (P.first_sample > 0) acc = mk(P.colors[3 * ci], P.colors[3 * ci + 1], P.colors[3 * ci + 2]);
    }

    uint32_t c_closest = 0, c_shadow = 0, c_draws = 0;
    uint32_t c_tests = 0;               // shadow-ray tests since the last flush into the workgroup's sum

    // ---- lane state ---------------------------------------------------------------------------
    enum : uint32_t { kNew = 0, kClosest = 1, kShadow = 2, kLights = 3 };
    PathCtl ctl{ kNew | 64u };
    V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f);     // the lane's own ray in flight: the path's, or the shadow ray (o = hit point)
    V3 thr = mk(1.f, 1.f, 1.f), rad = mk(0.f, 0.f, 0.f);
    Piece W;
    W.o = o;
    W.d = d;
    W.R = bvh_ray(s_hdr, o, d);
    W.cur = kWalkDone;
    W.sp = W.base = 0;
    W.far = 0.f;
    W.bnd = kNoIndex;
    W.own = (uint32_t)tid << 1;
    V3 nl = mk(0.f, 0.f, 1.f), ld = mk(0.f, 0.f, 0.f);
    float l_k = 0.f;
#if RT_OPT_WALK_CENSUS
    unsigned long long cen[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif

    for (;;) {
        const bool finished = ctl.st() == kNew && s >= s_end;
        if (__builtin_amdgcn_ballot_w64(!finished) == 0ull) break;

        // ---- T: everything the wavefront has to walk, by all its lanes ----
#if RT_OPT_WALK_CENSUS
        cen[8] += lane == 0 ? 1ull : 0ull;
        walk_shared(s_pairs, s_slots, s_index, s_stack, s_box, s_match, n_always, tid, P.walk_round, P.walk_take & 0xff, (P.walk_take >> 8) + 1, W, cen);
#else
        walk_shared(s_pairs, s_slots, s_index, s_stack, s_box, s_match, n_always, tid, P.walk_round, P.walk_take & 0xff, (P.walk_take >> 8) + 1, W, nullptr);
#endif

        // ---- S: every lane that has a pixel to go on with ----
        if (!finished) {
            bool path_done = false;
            int start = 0;                  // the ray this lane starts at the end of the phase: 0 none, 1 closest hit, 2 shadow
            if (ctl.st() == kShadow) {
                // ---- the shadow ray of light ctl.light() - 1 has its answer, .cl:297-301 ----
                const uint32_t first = (uint32_t)(s_box[tid] >> 32);
                const bool blocked = first < n;
                c_tests += blocked ? first + 1u : n;
                if ((int)c_tests < 0) {
                    atomicAdd(&s_stat[3], (unsigned long long)c_tests);
                    c_tests = 0u;
                }
                if (!blocked) {
                    const float4 lb = s_lightB[ctl.light() - 1u];
                    ld = add(ld, scale(mk(lb.x, lb.y, lb.z), l_k));
                }
                ctl.set_st(kLights);
            } else if (ctl.st() == kClosest) {
                c_closest += 1;
                const unsigned long long found = s_box[tid];
                const float t_hit = __uint_as_float((uint32_t)(found >> 32));
                if (!(t_hit < 1e20f)) {
                    path_done = true;                                              // miss, .cl:327-330
                } else {
                    const uint32_t id = (uint32_t)found;
                    const float4 ge = P.scene.geom[id];
                    const float4 em4 = P.scene.emis[id];
                    const float4 co4 = P.scene.colr[id];
                    const V3 em = mk(em4.x, em4.y, em4.z);
                    const V3 col = mk(co4.x, co4.y, co4.z);
                    const int refl = __float_as_int(em4.w);
                    const V3 hp = add(o, scale(d, t_hit));                         // .cl:338-340
                    const V3 nrm = unit(sub(hp, mk(ge.x, ge.y, ge.z)));            // .cl:345-347
                    const float dp = dot(nrm, d);
                    nl = scale(nrm, -1.f * cl_sign(dp));                           // .cl:354-355
                    if (!((em.x == 0.f) && (em.z == 0.f))) {                       // .cl:358-368
                        if (ctl.after_specular()) rad = add(rad, mul(thr, scale(em, fabsf(dp))));
                        path_done = true;
                    } else if (refl == RT_DIFF) {                                  // .cl:370-373
                        ctl.set_after_specular(false);
                        thr = mul(thr, col);
                        o = hp;
                        ld = mk(0.f, 0.f, 0.f);
                        ctl.first_light();
                        ctl.set_st(kLights);
                    } else {
                        // mirror / glass, .cl:413-489 (as in rt_trace.inc.h)
                        const V3 rfl = sub(d, scale(nrm, 2.f * dp));
                        ctl.set_after_specular(true);
                        if (refl == RT_SPEC) {
                            thr = mul(thr, col);
                            d = rfl;
                        } else {
                            const bool into = dp < 0.f;
                            const float ddn = -fabsf(dp);
                            const float nc = 1.f, nt = 1.52f;
                            float nnt = into ? nc / nt : nt / nc;
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
                                float pick = next_random(s0, s1);
                                c_draws += 1;
                                const bool take_rfl = pick < Pr;
                                const float wgt = rt_div(take_rfl ? Re : Tr, take_rfl ? Pr : 1.f - Pr);
                                thr = mul(scale(thr, wgt), col);
                                d = take_rfl ? rfl : td;
                            }
                        }
                        o = hp;
                        ctl.deeper();
                        if (ctl.depth() >= (uint32_t)kMaxDepth) path_done = true;  // .cl:320
                        else start = 1;
                    }
                }
            }
            // ---- next-event estimation, .cl:249-303, then the cosine-weighted bounce, .cl:383-411: one section for the two random
            //      numbers, the sine / cosine and the square root both begin with (as in rt_walk.inc.h) ----
            while (ctl.st() == kLights) {
                const bool bounce = ctl.light() == n_lights;
                const float f0 = __uint_as_float(next_random_word(s0, s1));        // first draw, in [2, 4)
                const float f1 = __uint_as_float(next_random_word(s0, s1));        // second draw
                c_draws += 2;
                const float turn = __builtin_fmaf(bounce ? f0 : f1, 0.5f, -1.0f);
                const float zc = 3.0f - f0;
                const float r2 = __builtin_fmaf(f1, 0.5f, -1.0f);
                const float under = bounce ? r2 : fmaxf(0.f, 1.f - zc * zc);
                const float root = rt_sqrt_unit(under);
                float sphi, cphi;
#if RT_FAST
                fm_sincos_turns(turn, sphi, cphi);
#else
                dm_sincosf_pos((2.f * RT_PI) * turn, sphi, cphi);
#endif
                if (bounce) {
                    rad = add(rad, mul(thr, ld));                                  // .cl:377-378
                    V3 w = nl;
                    V3 a = (fabsf(w.x) > .1f) ? mk(0.f, 1.f, 0.f) : mk(1.f, 0.f, 0.f);
                    V3 uu = unit(cross(a, w));
                    V3 vv = cross(w, uu);
                    V3 nd = add(scale(uu, cphi * root), scale(vv, sphi * root));
                    nd = add(nd, scale(w, rt_sqrt_unit(1 - r2)));
                    d = nd;
                    ctl.deeper();
                    ctl.set_st(kNew);                                              // (leaves the light loop)
                    if (ctl.depth() >= (uint32_t)kMaxDepth) path_done = true;
                    else start = 1;
                } else {
                    const float4 la = s_lightA[ctl.light()], lb = s_lightB[ctl.light()];
                    ctl.next_light();
                    const V3 us = mk(root * cphi, root * sphi, zc);                // .cl:203-213
                    const V3 on_light = add(scale(us, la.w), mk(la.x, la.y, la.z));
                    V3 sd = sub(on_light, o);
                    float len;
                    sd = scale(sd, sqrt_and_rcp(dot(sd, sd), len));
                    float wo = dot(sd, us);
                    const float wi = dot(sd, nl);
                    if (!(wo > 0.f) && wi > 0.f) {                                 // .cl:283-296: this side of the light, facing it
                        wo = -wo;
                        c_shadow += 1;
                        l_k = rt_div(lb.w * wi * wo, len * len);                   // .cl:297 (used only if nothing blocks)
                        d = sd;
                        W.far = len - RT_EPS;
                        ctl.set_st(kShadow);
                        start = 2;
                    }
                }
            }
            if (path_done) {
                // ---- running average, .cl:580-589 ----
                if (s == 0) {
                    acc = rad;
                } else {
                    float k1 = (float)s;
                    float k2 = k2_in_lds ? s_k2[s - P.first_sample] : rt_rcp((float)s + 1.f);
                    acc = mk((acc.x * k1 + rad.x) * k2, (acc.y * k1 + rad.y) * k2, (acc.z * k1 + rad.z) * k2);
                }
                s += 1;
                ctl.set_st(kNew);
                start = 0;
            }
            if (ctl.st() == kNew && start == 0 && s < s_end) {
                // ---- camera ray, .cl:494-549; the camera (12 floats) and 1/w, 1/h come from LDS, once per sample ----
                const float4 *cam_p = s_cam;
                asm volatile("; camera read here, once per sample" : "+v"(cam_p));
                const float4 c0 = cam_p[0], c1 = cam_p[1], c2 = cam_p[2], c3 = cam_p[3];
                const float inv_w = c3.z, inv_h = c3.w;
                const V3 cam_o = mk(c0.x, c0.y, c0.z), cam_d = mk(c0.w, c1.x, c1.y);
                const V3 cam_x = mk(c1.z, c1.w, c2.x), cam_y = mk(c2.y, c2.z, c2.w);
                float j1 = next_random_centred(s0, s1);
                float j2 = next_random_centred(s0, s1);
                c_draws += 2;
                float kcx = ((float)(xy & 0xffffu) + j1) * inv_w - 0.5f;
                float kcy = ((float)(xy >> 16) + j2) * inv_h - 0.5f;
                V3 rd = mk(cam_x.x * kcx + cam_y.x * kcy + cam_d.x, cam_x.y * kcx + cam_y.y * kcy + cam_d.y,
                           cam_x.z * kcx + cam_y.z * kcy + cam_d.z);
                o = add(scale(rd, 0.1f), cam_o);
                d = unit(rd);
                thr = mk(1.f, 1.f, 1.f);
                rad = mk(0.f, 0.f, 0.f);
                ctl.new_path();
                start = 1;
            }
            if (start != 0) {
                // ---- a new ray, closest hit (.cl:215-232) or shadow (.cl:234-247): the large spheres now, in scene order; what they
                //      give is the first entry of the ray's word; the tree in the trips to come ----
                const bool shadow = start == 2;
                float t = shadow ? W.far : 1e20f;
                uint32_t slot = 0, first = n_always;
                for (uint32_t i = 0; i < n_always; ++i) {
                    const HitPre p0 = hit_pre(s_slots[i], o, d);
                    if (wave_any_nonneg(p0.det)) {
                        const HitRoots h0 = hit_roots(p0);
                        const bool nearer = h0.hit & (h0.t < t);
                        first = (shadow & nearer & (first == n_always)) ? i : first;
                        const bool take = nearer & !shadow;
                        t = take ? h0.t : t;
                        slot = take ? i : slot;
                    }
                }
#if RT_OPT_WALK_CENSUS
                cen[9] += shadow ? (first < n_always ? first + 1u : n_always) : n_always;
#endif
                if (shadow) {
                    W.bnd = first < n_always ? (uint32_t)s_index[first] : n;
                    s_box[tid] = (unsigned long long)W.bnd << 32;
                } else {
                    W.far = t;
                    W.bnd = kNoIndex;
                    s_box[tid] = ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(t < 1e20f ? (uint32_t)s_index[slot] : kNoIndex);
                    ctl.set_st(kClosest);
                }
                W.o = o;
                W.d = d;
                W.R = bvh_ray(s_hdr, o, d);
                W.cur = root_ref;
                W.sp = 0;
                W.base = 0;
                W.own = ((uint32_t)tid << 1) | (shadow ? 1u : 0u);
            }
        }
    }
#if RT_OPT_WALK_CENSUS
    for (int k = 0; k < 10; ++k) {
        unsigned long long v = cen[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) atomicAdd(&P.counters[20 + k], v);
    }
#endif

    // ---- epilogue: as in rt_trace.inc.h ----
    const __attribute__((address_space(4))) LaunchParams *qp =
        (const __attribute__((address_space(4))) LaunchParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; epilogue arguments re-read" : "+s"(qp));
    const __attribute__((address_space(4))) LaunchParams &Q = *qp;
    const bool valid_e = s_end != Q.first_sample;
    if (valid_e && Q.n_samples > 0) {
        const int xe = (int)(xy & 0xffffu), ye = (int)(xy >> 16);
        int le = tile_by * kTileH + ((int)(threadIdx.x & 63u) >> 3);
        if (Q.deal) {
            const int rows_e = Q.deal_rows, bands = rows_e >> 3, ry_e = tile_by / bands, band = tile_by - ry_e * bands;
            le = ry_e * rows_e + (int)(Q.deal[(size_t)(ry_e * (int)gridDim.x + tile_bx) * (size_t)(kRegionW * rows_e) + (unsigned)(band * 256 + (int)threadIdx.x)] >> 5);
        }
        const size_t gid = (size_t)ye * (size_t)Q.w + (size_t)xe;
        const size_t ci = (size_t)(Q.h - ye - 1) * (size_t)Q.w + (size_t)xe;
        float *colors = Q.colors;
        colors[3 * ci] = acc.x;
        colors[3 * ci + 1] = acc.y;
        colors[3 * ci + 2] = acc.z;
        if (!Q.skip_pixels)
            Q.pixels[(size_t)le * (size_t)Q.w + (size_t)xe] =
                (uint32_t)(to_int(acc.x) | (to_int(acc.y) << 8) | (to_int(acc.z) << 16));
        *reinterpret_cast<uint2 *>(Q.seeds + 2 * gid) = make_uint2(s0, s1);
        uint16_t *pc = Q.pixel_cost;
        if (pc) {
            const uint32_t rays = c_closest + c_shadow;
            pc[(size_t)le * (size_t)Q.w + (size_t)xe] = (uint16_t)(rays < 65535u ? rays : 65535u);
        }
    }
    uint32_t n_done = valid_e ? (uint32_t)Q.n_samples : 0u;
    uint32_t t_samples = wave_sum(n_done);
    uint32_t t_closest = wave_sum(c_closest);
    uint32_t t_shadow = wave_sum(c_shadow);
    uint32_t t_draws = wave_sum(c_draws);
    unsigned long long tests64 = (unsigned long long)c_tests + (unsigned long long)c_closest * n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tests64 += __shfl_xor(tests64, off, 64);
    if (lane == 0) atomicMax(&s_tile_cost, (unsigned)(__builtin_amdgcn_s_memrealtime() - s_wave_t0[wave]));
    if (lane == 0) {
        atomicAdd(&s_stat[0], (unsigned long long)t_samples);
        atomicAdd(&s_stat[1], (unsigned long long)t_closest);
        atomicAdd(&s_stat[2], (unsigned long long)t_shadow);
        atomicAdd(&s_stat[3], tests64);
        atomicAdd(&s_stat[4], (unsigned long long)t_draws);
    }
    __syncthreads();
    if (tid == 5 && Q.tile_cost) Q.tile_cost[tile_id] = s_tile_cost;
    if (tid < 5) atomicAdd(&Q.stats[(block_linear % (unsigned)kStatReplicas) * 8u + (unsigned)tid], s_stat[tid]);
}
