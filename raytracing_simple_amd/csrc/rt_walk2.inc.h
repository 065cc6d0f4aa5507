// rt_walk2.inc.h -- the path-trace kernel for large scenes, second form (RT_OPT_WALK 3; 4 = the same with a census),
// included by rt_trace.inc.h in place of its own kernel body after rt_walk.inc.h's helpers (BvhRay, bvh_ray, bvh_misses).
//
// Same mapping, same arithmetic, same order of operations and random draws per pixel as rt_walk.inc.h -- one lane = one pixel,
// the ray's walk through the hierarchy is lane state that survives loop trips, shading happens when enough lanes wait -- and
// so the same frames, seeds and counters.  What differs is how the wavefront spends its instructions:
//
//   * the walk is ONE loop with wave-uniform control: each iteration is a pair step of the lanes that stand at a pair or a
//     leaf step of the lanes that hold a leaf, decided by two ballots (scalar branches), instead of per-lane nested loops
//     whose every level saves and restores the execution mask;
//   * a pair step is straight-line code: both boxes tested, the stack's top read WITH the pair (a pop then waits for one LDS
//     round trip, not two), the kept sibling written above the top whether it is kept or not (the stack has a spare level),
//     the pruning of shadow walks by scene index folded into one compare against a per-ray bound (`prune`);
//   * a leaf step updates the best hit with selects; what needs memory or is rare leaves the loop over the eight spheres:
//     the scene indices of a shadow ray's blockers are read once the leaf is done (one wait, not one per blocker), and an
//     exact tie in distance (the .scn loader doubles spheres) sends that lane through the reference's rule afterwards;
//   * the shade phase runs each piece of work ONCE for whoever needs it: the two random draws, the sine / cosine and the
//     square root that a light sample and a diffuse bounce both begin with are one section for both (a wavefront's lanes
//     are typically half back from a closest-hit walk -- light sample next -- and half back from a shadow walk -- bounce
//     next), and every new ray, closest-hit or shadow, starts in one place (always-list sweep, ray set-up).

struct Walk {
    uint32_t cur;           // what the lane looks at next: a pair, kBvhLeafRef | leaf, or kWalkDone
    int sp;                 // entries on its stack
    float far;              // closest hit: the best distance so far; shadow ray: its length (fixed)
    uint32_t idx;           // closest hit: scene index of the best (kWalkIndexOpen: not read yet); shadow ray: lowest blocking index so far
    uint32_t slot;          // closest hit: slot of the best
};

// One pair step of this lane (W.cur < kBvhLeafRef): rt_walk.inc.h walk_pairs' step without a branch.
RT_DEV void pair_step2(const float4 *s_pairs, uint16_t *my_stack, int stride, const BvhRay &R, bool shadow, Walk &W) {
    const float4 *pp = s_pairs + 4u * W.cur;
    const float4 A0 = pp[0], B0 = pp[1], A1 = pp[2], B1 = pp[3];
    const int below = W.sp > 0 ? W.sp - 1 : 0;
    const uint32_t top = my_stack[below * stride];
    float tn0, tn1;
    const bool out0 = bvh_misses(R, A0, B0, W.far, tn0), out1 = bvh_misses(R, A1, B1, W.far, tn1);
    // a shadow walk skips subtrees that hold only scene indices above its lowest blocker so far; a closest-hit walk never does
    const uint32_t prune = shadow ? W.idx : 0xffffffffu;
    const bool m0 = (int)out0 | (int)(__float_as_uint(B0.w) > prune), m1 = (int)out1 | (int)(__float_as_uint(B1.w) > prune);
    const uint32_t r0 = __float_as_uint(A0.w), r1 = __float_as_uint(A1.w);
    const bool both = !m0 & !m1, none = m0 & m1;
    const bool second_first = both ? (tn1 < tn0) : m0;
    const uint32_t near = second_first ? r1 : r0, far = second_first ? r0 : r1;
    my_stack[W.sp * stride] = (uint16_t)far;            // (dead unless `both`: the entry above the top)
    const uint32_t popped = W.sp > 0 ? top : kWalkDone;
    W.cur = none ? popped : near;
    W.sp = W.sp + (both ? 1 : 0) - ((none & (W.sp > 0)) ? 1 : 0);
}

// One leaf step of this lane (W.cur = kBvhLeafRef | leaf): the leaf's eight spheres through the reference's test.
RT_DEV void leaf_step2(const float4 *s_slots, const uint32_t *index, uint16_t *my_stack, int stride, uint32_t n_always, V3 o, V3 d,
                       bool shadow, Walk &W) {
    const uint32_t sl = n_always + (uint32_t)kBvhLeaf * (W.cur & (kBvhLeafRef - 1u));
    const int below = W.sp > 0 ? W.sp - 1 : 0;
    const uint32_t top = my_stack[below * stride];      // the next node, in flight under the sphere tests
    uint32_t blockers = 0u;
    // (two halves of four: all eight records in flight at once are 32 registers the kernel does not have)
#pragma unroll
    for (int half = 0; half < kBvhLeaf; half += 4) {
    float4 g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = s_slots[sl + (uint32_t)(half + k)];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        const int k = half + k4;
        const HitPre p = hit_pre(g[k4], o, d);
        if (wave_any_nonneg(p.det)) {
            const HitRoots hr = hit_roots(p);
            const bool nearer = hr.hit & (hr.t < W.far);
            blockers |= (shadow & nearer) ? (1u << k) : 0u;                    // shadow ray (.cl:234-247): a blocker; its index is read below
            const bool take = nearer & !shadow;                                  // closest hit (.cl:215-232): strictly nearer takes the slot
            // An exact tie in distance (the .scn loader doubles spheres: real) keeps the lower scene index, as the reference's loop
            // does: rare, so it is a scalar branch out of the straight line (rt_walk.inc.h walk_pairs has the rule as written).
            const bool tie = !shadow & hr.hit & (hr.t == W.far);
            if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {
                if (tie) {
                    const uint32_t ix = index[sl + (uint32_t)k];
                    const uint32_t have = W.idx == kWalkIndexOpen ? index[W.slot] : W.idx;
                    W.idx = have;
                    if (ix < have) {
                        W.slot = sl + (uint32_t)k;
                        W.idx = ix;
                    }
                }
            }
            W.far = take ? hr.t : W.far;
            W.slot = take ? sl + (uint32_t)k : W.slot;
            W.idx = take ? kWalkIndexOpen : W.idx;
        }
    }
    }
    if (__builtin_amdgcn_ballot_w64(blockers != 0u) != 0ull) {
        // the lowest scene index that blocks is the answer (and prunes what is left of the walk)
        while (__builtin_amdgcn_ballot_w64(blockers != 0u) != 0ull) {
            if (blockers != 0u) {
                const uint32_t k = (uint32_t)__builtin_ctz(blockers);
                blockers &= blockers - 1u;
                const uint32_t ix = index[sl + k];
                W.idx = ix < W.idx ? ix : W.idx;
            }
        }
    }
    W.cur = W.sp > 0 ? top : kWalkDone;
    W.sp = below;
}

// The same leaf step done by the WAVEFRONT for a few lanes.  Half of all leaf steps run with eight lanes or fewer -- the last
// walks of a trip, while the other lanes wait -- and each costs eight sphere tests one after the other.  Here the lanes that
// hold a leaf (`holders`, at most 8 per round) post their ray in a per-wavefront LDS mailbox, and each is served by a group of
// eight lanes: lane m of group q runs the reference's test of sphere m of ray q's leaf -- ONE test per lane, whoever's lane it
// is -- and the group combines with three DPP minima: the nearest hit's distance bits (a closest-hit ray), or the lowest
// blocking scene index (a shadow ray: the blockers read their index, the others do not).  The candidate and the rule are the
// per-lane step's; an exact tie in distance (two spheres of the leaf, or a sphere and the ray's best so far) sends the ray's
// own lane through that step instead.  Every lane of the wavefront is in the loop that calls this (no lane leaves the kernel's
// loop before the wavefront does), which is what lets a group count on its eight lanes.
constexpr int kCoopLeafRays = 8;
RT_DEV uint32_t group8_min(uint32_t key) {
    uint32_t k = key, t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)k, (int)k, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]: lane ^ 1
    k = t < k ? t : k;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]: lane ^ 2
    k = t < k ? t : k;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)k, (int)k, 0x141, 0xf, 0xf, false);     // row_half_mirror: lane -> 7 - lane of its eight
    k = t < k ? t : k;
    return k;
}
// Returns true for a holder that must still take the per-lane step.
RT_DEV bool leaf_step_coop(const float4 *s_slots, const uint32_t *index, uint16_t *my_stack, int stride, uint32_t n_always, V3 o, V3 d,
                           bool shadow, bool at_leaf, unsigned long long holders, float4 *mail, uint2 *res, Walk &W) {
    const int lane = (int)(threadIdx.x & 63u);
    const int q = lane >> 3, m = lane & 7;
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(holders >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)holders, 0u));
    const int n_hold = __popcll(holders);
    const uint32_t sl = n_always + (uint32_t)kBvhLeaf * (W.cur & (kBvhLeafRef - 1u));
    bool exact = false;                         // this holder must take the per-lane step (a tie)
    for (int first = 0; first < n_hold; first += kCoopLeafRays) {       // (wave-uniform: rounds of up to eight rays)
        const bool mine = at_leaf & (rank >= first) & (rank < first + kCoopLeafRays);
        if (mine) {
            mail[2 * (rank - first)] = make_float4(o.x, o.y, o.z, W.far);
            mail[2 * (rank - first) + 1] = make_float4(d.x, d.y, d.z, __uint_as_float(sl | (shadow ? 0x80000000u : 0u)));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool serve = q < n_hold - first;
        uint32_t key = 0xffffffffu, cand = 0u;
        bool tie = false;
        if (serve) {
            const float4 ra = mail[2 * q], rb = mail[2 * q + 1];
            const uint32_t tag = __float_as_uint(rb.w);
            cand = (tag & 0x7fffffffu) + (uint32_t)m;
            const bool sh = (tag >> 31) != 0u;
            const HitRoots hr = hit_roots(hit_pre(s_slots[cand], mk(ra.x, ra.y, ra.z), mk(rb.x, rb.y, rb.z)));
            const bool nearer = hr.hit & (hr.t < ra.w);
            tie = !sh & hr.hit & (hr.t == ra.w);
            if (sh) {
                if (nearer) key = index[cand];                                      // a blocker: its scene index
            } else {
                key = nearer ? __float_as_uint(hr.t) : 0xffffffffu;                 // (distances are positive: their bits order like they do)
            }
        }
        const uint32_t best = group8_min(key);
        const bool win = serve & (key == best) & (key != 0xffffffffu);
        const unsigned long long wins = __builtin_amdgcn_ballot_w64(win), ties = __builtin_amdgcn_ballot_w64(tie);
        const uint32_t my_wins = (uint32_t)(wins >> (8 * q)) & 0xffu, my_ties = (uint32_t)(ties >> (8 * q)) & 0xffu;
        const uint32_t need_exact = ((my_wins & (my_wins - 1u)) != 0u || my_ties != 0u) ? 0x80000000u : 0u;
        if (serve) {
            if (my_wins != 0u) {
                if ((my_wins & (0u - my_wins)) == (1u << m)) res[q] = make_uint2(best, cand | need_exact);      // the (first) winner reports
            } else if (m == 0) {
                res[q] = make_uint2(0xffffffffu, need_exact);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (mine) {
            const uint2 rr = res[rank - first];
            exact = (rr.y >> 31) != 0u;
            if (!exact) {
                if (shadow) {
                    W.idx = rr.x < W.idx ? rr.x : W.idx;
                } else if (rr.x != 0xffffffffu) {
                    W.far = __uint_as_float(rr.x);
                    W.slot = rr.y & 0x7fffffffu;
                    W.idx = kWalkIndexOpen;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (at_leaf & !exact) {
        const int below = W.sp > 0 ? W.sp - 1 : 0;
        W.cur = W.sp > 0 ? (uint32_t)my_stack[below * stride] : kWalkDone;
        W.sp = below;
    }
    return exact;
}

#undef RT_W2_COUNT
#undef RT_W2_CLOCK
#undef RT_W2_HIST

extern "C" __global__ void __launch_bounds__(64 * RT_OPT_WG_WAVES, RT_OPT_MINWAVES) RT_KERNEL_NAME(const LaunchParams P) {
    constexpr int kBlockThreads = 64 * RT_OPT_WG_WAVES;
    constexpr int kTileW = 8 * RT_OPT_WG_WAVES;
    extern __shared__ float4 lds[];
    const uint32_t n = P.scene.n_spheres;
    const uint32_t n_lights = P.scene.n_lights;
    const uint32_t n_always = P.bvh.n_always, n_slots = P.bvh.n_slots;
    // (the scene index of a slot is only read for a candidate that passes the test: from HBM / L2, not staged)
    const uint32_t *s_index = reinterpret_cast<const uint32_t *>(P.bvh.blob + bvh_index_at(n_slots));
    float4 *s_hdr = lds;
    const uint32_t n_pairs = P.bvh.n_leaves - 1u;
    const uint32_t stack_f4 = (P.bvh.stack_depth * (uint32_t)kBlockThreads * 2u + 15u) / 16u;
    const uint32_t root_ref = n_pairs ? P.bvh.root : kBvhLeafRef;
#if RT_OPT_GLOBAL_TABLES
    const float4 *s_pairs = P.bvh.blob + bvh_pairs_at(n_slots);
    const float4 *s_slots = P.bvh.blob + bvh_slots_at();
    uint16_t *s_stack = reinterpret_cast<uint16_t *>(s_hdr + 2);
    const float4 *s_lightA = P.scene.lightA, *s_lightB = P.scene.lightB;
    float4 *s_emis = s_hdr + 2 + stack_f4;        // (never read: the host keeps mat_in_lds off)
    float4 *s_colr = s_emis;
#else
    // staged: hdr | pairs | slots | one stack of P.bvh.stack_depth u16 per lane ([level][lane]) | lights | materials
    float4 *s_pairs = s_hdr + 2;
    float4 *s_slots = s_pairs + 4 * n_pairs;
    uint16_t *s_stack = reinterpret_cast<uint16_t *>(s_slots + n_slots);
    float4 *s_lightA = s_slots + n_slots + stack_f4;     // {centre, radius}
    float4 *s_lightB = s_lightA + n_lights;       // {emission, 4*pi*radius^2}
    float4 *s_emis = s_lightB + n_lights;         // {emission, bits(refl)}   (if mat_in_lds)
    float4 *s_colr = s_emis + n;                  // {colour, radius}
#endif
    float *s_k2 = reinterpret_cast<float *>(P.mat_in_lds ? s_colr + n : s_emis);
    const bool k2_in_lds = P.n_samples <= kMaxK2Table;

    const int tid = threadIdx.x;
    __shared__ unsigned long long s_stat[5];
    __shared__ unsigned s_tile_cost;
    __shared__ float4 s_cam[4];         // orig, dir | x, y | 1/w, 1/h
    if (tid < 5) s_stat[tid] = 0;
    if (tid == 5) s_tile_cost = 0u;
    if (tid == 6) {
        s_cam[0] = make_float4(P.cam.orig.x, P.cam.orig.y, P.cam.orig.z, P.cam.dir.x);
        s_cam[1] = make_float4(P.cam.dir.y, P.cam.dir.z, P.cam.x.x, P.cam.x.y);
        s_cam[2] = make_float4(P.cam.x.z, P.cam.y.x, P.cam.y.y, P.cam.y.z);
        s_cam[3] = make_float4(0.f, 0.f, P.inv_w, P.inv_h);
    }
    if (tid < 2) s_hdr[tid] = P.bvh.blob[tid];
#if !RT_OPT_GLOBAL_TABLES
    {
        const float4 *g_pairs = P.bvh.blob + bvh_pairs_at(n_slots);
        const float4 *g_slots = P.bvh.blob + bvh_slots_at();
        for (uint32_t i = tid; i < 4u * n_pairs; i += kBlockThreads) s_pairs[i] = g_pairs[i];
        for (uint32_t i = tid; i < n_slots; i += kBlockThreads) s_slots[i] = g_slots[i];
    }
    for (uint32_t i = tid; i < n_lights; i += kBlockThreads) {
        s_lightA[i] = P.scene.lightA[i];
        s_lightB[i] = P.scene.lightB[i];
    }
#endif
    if (P.mat_in_lds) {
        for (uint32_t i = tid; i < n; i += kBlockThreads) {
            s_emis[i] = P.scene.emis[i];
            s_colr[i] = P.scene.colr[i];
        }
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
    // through the loop the pixel's place is ONE register, x | y << 16 (the camera ray needs both per sample; images stop at 65535
    // either way); the local row and the validity are formed again after the loop (as in rt_trace.inc.h)
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
    uint32_t c_tests = 0;               // shadow-ray tests since the last flush into the workgroup's sum (one register, not two: below)

    // ---- lane state ---------------------------------------------------------------------------
    enum { kNew = 0, kClosest = 1, kShadow = 2, kLights = 3 };
    int st = kNew;
    V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f);     // the ray in flight: the path's, or the shadow ray (o = hit point)
    V3 thr = mk(1.f, 1.f, 1.f), rad = mk(0.f, 0.f, 0.f);
    int depth = 0;
    bool after_specular = true;
    Walk W{ kWalkDone, 0, 0.f, 0xffffffffu, 0u };
    BvhRay R = bvh_ray(s_hdr, o, d);
    // a diffuse hit being lit: its normal, the light sum, the light in flight and what it adds if unblocked
    V3 nl = mk(0.f, 0.f, 1.f), ld = mk(0.f, 0.f, 0.f);
    uint32_t lj = 0;
    float l_k = 0.f;
    uint16_t *my_stack = s_stack + tid;
#if RT_OPT_WALK == 4
    // census instance.  cen[0/1] pair steps (two box tests each) per wavefront / per lane, [2/3] leaf steps (kBvhLeaf sphere tests
    // each), [4/5] shade phases, [6/7] clock ticks in the walk / in shading, [8] loop trips, [9] sphere tests of the always-list
    // sweeps -> counters[20..29] (the layout of rt_walk.inc.h's census); hist[0..3] leaf steps with 1-8 / 9-16 / 17-32 / 33-64
    // lanes, hist[4..7] pair steps likewise, hist[8/9] clock ticks in pair / leaf steps -> counters[8..17]
    unsigned long long cen[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long hist[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#define RT_W2_COUNT(k, mask)                                                                     \
    do {                                                                                         \
        if (lane == __ffsll((long long)(mask)) - 1) cen[k] += 1ull;                              \
        cen[(k) + 1] += 1ull;                                                                    \
    } while (0)
#define RT_W2_HIST(base, mask)                                                                   \
    do {                                                                                         \
        const int n_ = __popcll(mask);                                                           \
        if (lane == __ffsll((long long)(mask)) - 1) hist[(base) + (n_ <= 8 ? 0 : (n_ <= 16 ? 1 : (n_ <= 32 ? 2 : 3)))] += 1ull; \
    } while (0)
#define RT_W2_CLOCK(arr, k, t0)                                                                  \
    do {                                                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                            \
        const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true);                       \
        if (lane == __ffsll((long long)act_) - 1) arr[k] += now_ - (t0);                         \
    } while (0)
#else
#define RT_W2_COUNT(k, mask)
#define RT_W2_HIST(base, mask)
#define RT_W2_CLOCK(arr, k, t0)
#endif

    __shared__ __attribute__((aligned(16))) float4 s_mail[RT_OPT_WG_WAVES][2 * kCoopLeafRays];
    __shared__ uint2 s_res[RT_OPT_WG_WAVES][kCoopLeafRays];
    for (;;) {
        // (a lane that has rendered its samples stays in the loop, idle, until its wavefront has: the cooperative leaf step
        // counts on all 64 lanes, and the loop's exit is then a scalar branch)
        const bool finished = st == kNew && s >= s_end;
        if (__builtin_amdgcn_ballot_w64(!finished) == 0ull) break;
#if RT_OPT_WALK == 4
        cen[8] += (lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) ? 1ull : 0ull;
        const unsigned long long t_trip = __builtin_amdgcn_s_memtime();
#endif

        // ---- T: every walk in flight advances, up to P.walk_steps pair steps; at most P.walk_round of them in a row while
        //      some lane holds a leaf.  Control is the wavefront's: two ballots per step, scalar branches ----
#if RT_OPT_WALK == 9
        if (W.cur != kWalkDone)
            walk_pairs_d(s_pairs, s_slots, s_index, my_stack, kBlockThreads, n_always, o, d, R, st == kShadow, P.walk_round, W.cur, W.sp, W.far, W.idx, W.slot);
#elif RT_OPT_WALK == 7 || RT_OPT_WALK == 8
        if (W.cur != kWalkDone)
            walk_pairs_c(s_pairs, s_slots, s_index, my_stack, kBlockThreads, n_always, o, d, R, st == kShadow, P.walk_steps, P.walk_round, W.cur, W.sp,
                         W.far, W.idx, W.slot);
#elif RT_OPT_WALK == 6
        if (W.cur != kWalkDone)
            walk_pairs_b(s_pairs, s_slots, s_index, my_stack, kBlockThreads, n_always, o, d, R, st == kShadow, P.walk_steps, P.walk_round, W.cur, W.sp,
                         W.far, W.idx, W.slot);
#elif RT_OPT_WALK == 5
        // (A/B: this form's shade phase over rt_walk.inc.h's walk loop)
        if (W.cur != kWalkDone)
            walk_pairs(s_pairs, s_slots, s_index, my_stack, kBlockThreads, n_always, o, d, R, st == kShadow, P.walk_steps, P.walk_round, W.cur, W.sp,
                       W.far, W.idx, W.slot, nullptr);
#else
        {
            // (the two counters are the wavefront's: read through the first lane so that the compiler keeps them, and the
            // branches on them, scalar)
            int budget = __builtin_amdgcn_readfirstlane(P.walk_steps), round = __builtin_amdgcn_readfirstlane(P.walk_round);
            const bool shadow = st == kShadow;
            for (;;) {
                const bool at_pair = W.cur < kBvhLeafRef;
                const bool at_leaf = (W.cur >= kBvhLeafRef) & (W.cur != kWalkDone);
                const unsigned long long bp = __builtin_amdgcn_ballot_w64(at_pair), bl = __builtin_amdgcn_ballot_w64(at_leaf);
                if (bp != 0ull && budget > 0 && (round > 0 || bl == 0ull)) {
                    budget = __builtin_amdgcn_readfirstlane(budget - 1);
                    round = __builtin_amdgcn_readfirstlane(round - 1);
#if RT_OPT_WALK == 4
                    const unsigned long long t_p = __builtin_amdgcn_s_memtime();
#endif
                    if (at_pair) {
                        RT_W2_COUNT(0, bp);
                        RT_W2_HIST(4, bp);
                        pair_step2(s_pairs, my_stack, kBlockThreads, R, shadow, W);
                    }
                    RT_W2_CLOCK(hist, 8, t_p);
                } else if (bl != 0ull) {
                    round = __builtin_amdgcn_readfirstlane(P.walk_round);
#if RT_OPT_WALK == 4
                    const unsigned long long t_l = __builtin_amdgcn_s_memtime();
#endif
                    if (at_leaf) {
                        RT_W2_COUNT(2, bl);
                        RT_W2_HIST(0, bl);
                    }
                    bool by_lane = at_leaf;
                    if (__popcll(bl) <= P.walk_tail)
                        by_lane = leaf_step_coop(s_slots, s_index, my_stack, kBlockThreads, n_always, o, d, shadow, at_leaf, bl, s_mail[wave], s_res[wave], W);
                    if (__builtin_amdgcn_ballot_w64(by_lane) != 0ull) {
                        if (by_lane) leaf_step2(s_slots, s_index, my_stack, kBlockThreads, n_always, o, d, shadow, W);
                    }
                    RT_W2_CLOCK(hist, 9, t_l);
                } else {
                    break;
                }
            }
        }
#endif

#if RT_OPT_WALK == 4
        RT_W2_CLOCK(cen, 6, t_trip);
        const unsigned long long t_s = __builtin_amdgcn_s_memtime();
#endif
        // ---- S: lanes whose walk has ended, once enough of them wait ----
        const bool ready = (W.cur == kWalkDone) & !finished;
        const unsigned long long br = __builtin_amdgcn_ballot_w64(ready);
        const unsigned long long bw = __builtin_amdgcn_ballot_w64(W.cur != kWalkDone);
        const bool go = (__popcll(br) >= P.regen_gate) || (bw == 0ull);
        if (ready && go) {
            RT_W2_COUNT(4, __builtin_amdgcn_ballot_w64(true));
            bool path_done = false;
            int start = 0;                  // the ray this lane starts at the end of the phase: 0 none, 1 closest hit, 2 shadow
            if (st == kShadow) {
                // ---- the shadow ray of light lj - 1 has its answer, .cl:297-301 ----
                const bool blocked = W.idx < n;
                c_tests += blocked ? W.idx + 1u : n;
                if ((int)c_tests < 0) {                                             // (rare: the 64-bit sum lives in LDS, the lane keeps 31 bits of it)
                    atomicAdd(&s_stat[3], (unsigned long long)c_tests);
                    c_tests = 0u;
                }
                if (!blocked) {
                    const float4 lb = s_lightB[lj - 1u];
                    ld = add(ld, scale(mk(lb.x, lb.y, lb.z), l_k));
                }
                st = kLights;
            } else if (st == kClosest) {
                c_closest += 1;
                if (!(W.far < 1e20f)) {
                    path_done = true;                                              // miss, .cl:327-330
                } else {
                    const float4 ge = s_slots[W.slot];
                    float4 em4, co4;
#if RT_OPT_WALK >= 8
                    // the hit sphere's material by the SLOT the walk ended on (the blob's material sections are in slot order):
                    // one round trip to L2, where the scene index first and the record by index after it were two
                    em4 = P.bvh.blob[P.bvh.emis_at + W.slot];
                    co4 = P.bvh.blob[P.bvh.emis_at + n_slots + W.slot];
#else
                    const uint32_t id = W.idx == kWalkIndexOpen ? s_index[W.slot] : W.idx;
                    if (P.mat_in_lds) {
                        em4 = s_emis[id];
                        co4 = s_colr[id];
                        asm volatile("; materials from LDS" : "+v"(em4.x));
                    } else {
                        em4 = P.scene.emis[id];
                        co4 = P.scene.colr[id];
                    }
#endif
                    const V3 em = mk(em4.x, em4.y, em4.z);
                    const V3 col = mk(co4.x, co4.y, co4.z);
                    const int refl = __float_as_int(em4.w);
                    const V3 hp = add(o, scale(d, W.far));                         // .cl:338-340
                    const V3 nrm = unit(sub(hp, mk(ge.x, ge.y, ge.z)));            // .cl:345-347
                    const float dp = dot(nrm, d);
                    nl = scale(nrm, -1.f * cl_sign(dp));                           // .cl:354-355
                    if (!((em.x == 0.f) && (em.z == 0.f))) {                       // .cl:358-368
                        if (after_specular) rad = add(rad, mul(thr, scale(em, fabsf(dp))));
                        path_done = true;
                    } else if (refl == RT_DIFF) {                                  // .cl:370-373
                        after_specular = false;
                        thr = mul(thr, col);
                        o = hp;
                        ld = mk(0.f, 0.f, 0.f);
                        lj = 0;
                        st = kLights;
                    } else {
                        // mirror / glass, .cl:413-489 (as in rt_trace.inc.h)
                        const V3 rfl = sub(d, scale(nrm, 2.f * dp));
                        after_specular = true;
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
                        depth += 1;
                        if (depth >= kMaxDepth) path_done = true;                  // .cl:320
                        else start = 1;
                    }
                }
            }
            // ---- next-event estimation, .cl:249-303 (the lights one by one, each with its two draws), then the cosine-weighted
            //      bounce, .cl:383-411 (two draws as well).  Either begins with two random numbers, the sine and cosine of 2 pi
            //      times one of them and the square root of a value formed from the other: that part is ONE section for the lanes
            //      about to sample a light and the lanes about to bounce.  Per pixel the operations and their order are the
            //      reference's (sample_light of rt_trace.inc.h and its bounce, term for term) ----
            while (st == kLights) {
                const bool bounce = lj == n_lights;
                const float f0 = __uint_as_float(next_random_word(s0, s1));        // first draw, in [2, 4)
                const float f1 = __uint_as_float(next_random_word(s0, s1));        // second draw
                c_draws += 2;
                // light: u1 -> z = 1 - 2 u1 = 3 - f0 (.cl:204), u2 -> phi = 2 pi u2 (.cl:208); bounce: r1 = 2 pi u (.cl:384), r2 (.cl:385)
                const float turn = __builtin_fmaf(bounce ? f0 : f1, 0.5f, -1.0f);
                const float zc = 3.0f - f0;
                const float r2 = __builtin_fmaf(f1, 0.5f, -1.0f);
                const float under = bounce ? r2 : fmaxf(0.f, 1.f - zc * zc);
                const float root = rt_sqrt_unit(under);                            // bounce: r2s = sqrt(r2); light: sqrt(max(0, 1 - z z))
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
                    depth += 1;
                    st = kNew;                                                     // (leaves the light loop)
                    if (depth >= kMaxDepth) path_done = true;
                    else start = 1;
                } else {
                    const float4 la = s_lightA[lj], lb = s_lightB[lj];
                    lj += 1u;
                    const V3 us = mk(root * cphi, root * sphi, zc);                // .cl:203-213
                    const V3 on_light = add(scale(us, la.w), mk(la.x, la.y, la.z));
                    V3 sd = sub(on_light, o);
                    float len;
                    sd = scale(sd, sqrt_and_rcp(dot(sd, sd), len));
                    float wo = dot(sd, us);
                    const float wi = dot(sd, nl);
                    if (!(wo > 0.f) && wi > 0.f) {                                 // .cl:283-296: this side of the light, facing it
                        wo = -wo;
                        // ---- shadow ray, any hit, .cl:234-247: the large spheres at the end of this phase, the tree in the trips to come ----
                        c_shadow += 1;
                        l_k = rt_div(lb.w * wi * wo, len * len);                   // .cl:297 (used only if nothing blocks)
                        d = sd;
                        W.far = len - RT_EPS;
                        st = kShadow;
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
                st = kNew;
                start = 0;
            }
            if (st == kNew && start == 0 && s < s_end) {
                {
                    // ---- camera ray, .cl:494-549 (a finished path's next sample; the first sample of the launch); the camera
                    //      (12 floats) and 1/w, 1/h come from LDS, once per sample ----
                    const float4 *cam_p = s_cam;
                    asm volatile("; camera read here, once per sample" : "+v"(cam_p));       // (not hoisted out of the loop into registers that are then spilled)
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
                    depth = 0;
                    after_specular = true;
                    start = 1;
                }
            }
            if (start != 0) {
                // ---- a new ray, closest hit (.cl:215-232) or shadow (.cl:234-247): the large spheres now, in scene order, the
                //      tree in the trips to come.  A shadow ray keeps the first of them that blocks, a closest-hit ray the nearest ----
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
#if RT_OPT_WALK == 4
                cen[9] += shadow ? (first < n_always ? first + 1u : n_always) : n_always;
#endif
                if (shadow) {
                    W.idx = first < n_always ? s_index[first] : n;
                } else {
                    W.far = t;
                    W.slot = slot;
                    W.idx = (t < 1e20f) ? kWalkIndexOpen : 0xffffffffu;           // (the always-list winner's index is read if it stays the winner)
                    st = kClosest;
                }
                R = bvh_ray(s_hdr, o, d);
                W.cur = root_ref;
                W.sp = 0;
            }
        }
#if RT_OPT_WALK == 4
        RT_W2_CLOCK(cen, 7, t_s);
#endif
    }
#if RT_OPT_WALK == 4
    for (int k = 0; k < 10; ++k) {
        unsigned long long v = cen[k], h = hist[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v += __shfl_xor(v, off, 64);
            h += __shfl_xor(h, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&P.counters[20 + k], v);
            atomicAdd(&P.counters[8 + k], h);
        }
    }
#endif

    // ---- epilogue: as in rt_trace.inc.h ----
    const __attribute__((address_space(4))) LaunchParams *qp =
        (const __attribute__((address_space(4))) LaunchParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; epilogue arguments re-read" : "+s"(qp));
    const __attribute__((address_space(4))) LaunchParams &Q = *qp;
    const bool valid_e = s_end != Q.first_sample;       // (s_end was first_sample + n_samples for the lanes that own a pixel)
    if (valid_e && Q.n_samples > 0) {
        const int xe = (int)(xy & 0xffffu), ye = (int)(xy >> 16);
        int le = tile_by * kTileH + ((int)(threadIdx.x & 63u) >> 3);
        if (Q.deal) {                                                      // (the local row of a dealt pixel: read again, not kept)
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
