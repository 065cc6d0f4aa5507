// rt_walk.inc.h -- the path-trace kernel for large scenes (RT_OPT_WALK), included by rt_trace.inc.h in place of its own
// kernel body; everything above it there (vector helpers, RNG, the sphere test, the sweeps, sample_light) is shared.
//
// The small spheres hang in a bounding-volume hierarchy (rt_device.h BvhTables, built by rt_bvh.hip) that each lane walks
// for its own ray; the few large ones are swept by every ray as before.  The hierarchy only selects candidates -- every
// candidate goes through the reference's test, and the winner is chosen by the reference's rule -- so frames and counters
// equal the plain sweep's.
//
// Same mapping (one lane = one pixel, the spp loop and the path state in registers), same arithmetic, same order of random
// draws per pixel -- but the lanes of a wavefront are decoupled once more: a ray's walk is lane state (where it is in the
// tree, its stack, the bound, the best so far) that survives loop trips:
//
//   T  every lane with a walk in flight takes up to P.walk_steps steps of it (walk_pairs below; closest-hit and shadow
//      rays run the same loop, they differ in how a candidate updates the state);
//   S  lanes whose walk has ended (or that have no ray) do what comes next for them -- process the hit, sample
//      the next light and start its shadow ray, add the light's contribution, bounce, finish the sample, start
//      a camera ray -- once P.regen_gate of them are waiting or nobody is walking, and then join T again.
//
// Nothing here depends on which trip a lane does what: per pixel the sequence of operations and random draws is the
// reference's.  RT_OPT_WALK 2 is the census instance of the diagnostics build (steps executed, lanes taking part, clock
// shares: counters[20..28]); with RT_OPT_GLOBAL_TABLES the pairs and slots are read where they lie in HBM / L2.
// Earlier forms (depth-first nodes with skip links, walked per call or as lane state) were measured and dropped:
// DESIGN.md section 5 names the commits.

// ---- walking the hierarchy ---------------------------------------------------------------------------
// A sphere can only matter to a ray if the reference's test (hit_pre / hit_roots above, binary32, rounded after
// every operation) returns a distance t for it, EPSILON < t <= t_max.  Where is X = o + t d then?  With op = fl(p - o),
// OP = |op|, B = op.d exactly, b = B + db the computed dot product (|db| <= 3u OP, u = 2^-24), det = b^2 - OP^2 + r^2 + e
// the computed discriminant (|e| <= 8u M^2, M^2 = OP^2 + r^2: one rounded square, a rounded three-term dot product,
// the rounded r*r, two rounded sums), sq = sqrt(det)(1 + th), |th| <= u, and tau = b -+ sq before its own rounding:
//     |X - p|^2 = tau^2 dd - 2 tau B + OP^2 = (sq^2 - b^2 + OP^2) + tau^2 (dd - 1) + 2 tau db
//               = r^2 + e + 2 th det + tau^2 (dd - 1) + 2 tau db,      |tau| <= OP + |r|, tau^2 <= 2 M^2,
// so |X - p|^2 <= r^2 + (19u + 2 |dd - 1|) M^2 <= r^2 + eps with eps := (64u + 4 |dd - 1|) M^2: X lies within
// |r| + min(sqrt(eps), eps / 2|r|) of the centre -- inside the sphere's box grown by that `pad` -- at a ray parameter
// in (0, t_max].  (The rounding of t itself and of p - o move X by at most 3u (OP + |o|); together with the slab
// arithmetic's own rounding, bvh_misses below, that is an eighth of the linear term 64u (|o| + OP + |r|) of the pad.)  The walk therefore tests each node's box, grown by
// `pad`, against the stretch [-pad, t_max + pad] of the ray, with OP bounded by the distance to the far side of the
// root box and |r| by the largest radius in the tree; the slab arithmetic's own rounding is inside the pad's linear
// term (bvh_misses below).  A lane whose direction is not a unit vector to within 10^-3, or not finite, gets an
// infinite pad: it visits everything, like the plain sweep.  Comparisons are written so that NaN means "visit".
struct BvhRay {
    V3 clo, chi, inv;      // 1 / direction and -(origin +- pad) / direction: a slab distance is one fused multiply-add
    float tback;           // how far behind the origin / beyond the current best a box still counts
};
RT_DEV BvhRay bvh_ray(const float4 *s_hdr, V3 o, V3 d) {
    const float4 h0 = s_hdr[0], h1 = s_hdr[1];
    const float u = 0x1p-24f, inf = __builtin_inff();
    const float dd = d.x * d.x + d.y * d.y + d.z * d.z;
    const V3 oc = sub(mk(h0.x, h0.y, h0.z), o);
    const float far = (__builtin_amdgcn_sqrtf(dot(oc, oc)) + h0.w) * 1.01f;     // >= |p - o| of every sphere in the tree
    const float skew = fabsf(dd - 1.f);
    const float eps = (64.f * u + 4.f * skew) * (far * far + h1.y * h1.y);
    float pad = fminf(__builtin_amdgcn_sqrtf(eps), eps * h1.z);                 // sqrt(r^2 + eps) - r, from above
    pad = pad * 1.01f + 64.f * u * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + far + h1.y);
    float tback = pad + 1e-6f * far;
    // (the products below must stay finite: origins and trees beyond 10^18 are walked in full as well)
    const bool sane = (skew < 1e-3f) & (fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + far < 1e18f);    // false for NaN
    pad = sane ? pad : inf;
    tback = sane ? tback : inf;
    BvhRay R;
    // A direction component of (nearly) zero -- cosine-weighted bounces off an axis-aligned normal produce exact
    // zeros a dozen times per frame -- is taken as +-10^-18: the ray then misses its true line by 10^-18 per unit of
    // length, and 1 / d and the products with it stay finite (an infinite 1 / d would turn both slab distances of
    // an axis into the same infinity, or into NaN, whichever side of the slab the origin is on).
    const float tiny = 1e-18f;
    R.inv = mk(__builtin_amdgcn_rcpf(__builtin_copysignf(fmaxf(fabsf(d.x), tiny), d.x)),
               __builtin_amdgcn_rcpf(__builtin_copysignf(fmaxf(fabsf(d.y), tiny), d.y)),
               __builtin_amdgcn_rcpf(__builtin_copysignf(fmaxf(fabsf(d.z), tiny), d.z)));
    R.clo = mk(-(o.x + pad) * R.inv.x, -(o.y + pad) * R.inv.y, -(o.z + pad) * R.inv.z);
    R.chi = mk(-(o.x - pad) * R.inv.x, -(o.y - pad) * R.inv.y, -(o.z - pad) * R.inv.z);
    R.tback = tback;
    return R;
}
// True when the ray stretch [-tback, t_far + tback] misses the grown box for certain.  A slab distance is
// fma(plane, 1/d, -(o +- pad)/d): the plane this computed distance really belongs to -- o +- pad + t d, exactly -- lies
// within  4u |plane - o| + 2u |o|  of the box's (one ulp of v_rcp_f32, the rounded shifted origin, its rounded product,
// the fused operation's own rounding), which the pad's linear term covers with the rest (above); so the point X of
// the derivation above, which is inside the grown box by that margin, is between the computed planes on every axis and
// its parameter inside [tn, tf].  Minimum and maximum drop NaN operands (a direction component of 0 against a plane
// through the origin): that axis then does not constrain.
RT_DEV bool bvh_misses(const BvhRay &R, float4 A, float4 B, float t_far, float &t_near) {
    const float x0 = __builtin_fmaf(A.x, R.inv.x, R.clo.x), x1 = __builtin_fmaf(B.x, R.inv.x, R.chi.x);
    const float y0 = __builtin_fmaf(A.y, R.inv.y, R.clo.y), y1 = __builtin_fmaf(B.y, R.inv.y, R.chi.y);
    const float z0 = __builtin_fmaf(A.z, R.inv.z, R.clo.z), z1 = __builtin_fmaf(B.z, R.inv.z, R.chi.z);
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), -R.tback));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t_far + R.tback));
    t_near = tn;                        // where the ray enters the grown box (an ordering hint, nothing more)
    return tn > tf;
}
RT_DEV bool bvh_misses(const BvhRay &R, float4 A, float4 B, float t_far) {
    float unused;
    return bvh_misses(R, A, B, t_far, unused);
}

// ---- the walk over sibling pairs (rt_device.h BvhTables `pairs`), nearer child first ----
// One step loads a pair (64 bytes), tests both boxes against the ray's stretch, goes on with the nearer of the
// children that are hit and keeps the other on the lane's stack (16 bits per entry, [level][lane]); with neither hit
// it takes the last kept one.  Which child comes first only decides how soon the bound shrinks: every sphere whose
// chain of boxes the ray meets is still tested, so the result is the same set of candidates run through the same rule.
// Shadow rays look for the LOWEST blocking scene index (that is what .cl:234-247 returns at): a subtree that only
// holds higher indices than the best so far is skipped.  The walk's place (cur, sp) and its result so far (w_far,
// w_idx, w_slot) are the caller's: `budget` pair steps at most per call, the rest next time.  At most `round_len`
// pair steps are taken in a row before the leaf step of the lanes that hold a leaf -- a lane that is still looking
// goes on looking in the next round instead of keeping the others waiting (the number of steps to the next leaf has
// a long tail).  cen (census instances only): [0] pair steps of the wavefront, [1] of this lane, [2]/[3] leaf steps.
constexpr uint32_t kWalkDone = 0xffffffffu;
constexpr uint32_t kWalkIndexOpen = 0xfffffffeu;        // closest-hit walks: the best slot's scene index has not been read yet
RT_DEV void walk_pairs(const float4 *s_pairs, const float4 *s_slots, const uint32_t *index, uint16_t *my_stack, int stack_stride,
                       uint32_t n_always, V3 o, V3 d, const BvhRay &R, bool shadow, int budget, int round_len, uint32_t &cur,
                       int &sp, float &w_far, uint32_t &w_idx, uint32_t &w_slot, unsigned long long *cen) {
    const int lane_ = threadIdx.x & 63;
    while (cur != kWalkDone && budget > 0) {
        for (int round = round_len; cur < kBvhLeafRef && budget > 0 && round > 0; --round) {
            budget -= 1;
            if (cen) {
                const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true);
                if (lane_ == __ffsll((long long)act_) - 1) cen[0] += 1ull;
                cen[1] += 1ull;
            }
            const float4 *pp = s_pairs + 4u * cur;
            const float4 A0 = pp[0], B0 = pp[1], A1 = pp[2], B1 = pp[3];
            float tn0, tn1;
            const bool out0 = bvh_misses(R, A0, B0, w_far, tn0), out1 = bvh_misses(R, A1, B1, w_far, tn1);
            const bool m0 = out0 || (shadow && __float_as_uint(B0.w) > w_idx);
            const bool m1 = out1 || (shadow && __float_as_uint(B1.w) > w_idx);
            const uint32_t r0 = __float_as_uint(A0.w), r1 = __float_as_uint(A1.w);
            const bool both = !m0 & !m1, none = m0 & m1;
            const bool second_first = both ? (tn1 < tn0) : m0;
            const uint32_t near = second_first ? r1 : r0, far = second_first ? r0 : r1;
            if (both) {
                my_stack[sp * stack_stride] = (uint16_t)far;
                sp += 1;
            }
            if (none) {
                sp -= 1;
                cur = sp >= 0 ? (uint32_t)my_stack[sp * stack_stride] : kWalkDone;
                sp = sp < 0 ? 0 : sp;
            } else {
                cur = near;
            }
        }
        if (cur != kWalkDone && cur >= kBvhLeafRef) {
            if (cen) {
                const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true);
                if (lane_ == __ffsll((long long)act_) - 1) cen[2] += 1ull;
                cen[3] += 1ull;
            }
            const uint32_t sl = n_always + (uint32_t)kBvhLeaf * (cur & (kBvhLeafRef - 1u));
            // (two halves of four: the discriminants of all eight at once are registers the kernel then spills elsewhere)
#pragma unroll
            for (int half = 0; half < kBvhLeaf; half += 4) {
            HitPre p[4];
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) p[k4] = hit_pre(s_slots[sl + (uint32_t)(half + k4)], o, d);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int k = half + k4;
                if (p[k4].det >= 0.f) {
                    const HitRoots hr = hit_roots(p[k4]);
                    // closest hit (.cl:215-232): a smaller distance, or the same from a lower scene index;
                    // shadow ray (.cl:234-247): the lowest scene index that blocks
                    // The scene index of a slot lies in HBM / L2.  A shadow ray needs it for every blocker (the lowest one
                    // is the answer and prunes subtrees); a closest-hit ray only to break an exact tie -- a strictly
                    // nearer hit just takes the slot, and its index is read once, when the walk is over (kWalkIndexOpen).
                    if (shadow) {
                        if (hr.hit && hr.t < w_far) {
                            const uint32_t ix = index[sl + k];
                            w_idx = ix < w_idx ? ix : w_idx;
                        }
                    } else if (hr.hit) {
                        if (hr.t < w_far) {
                            w_far = hr.t;
                            w_slot = sl + (uint32_t)k;
                            w_idx = kWalkIndexOpen;
                        } else if (hr.t == w_far) {                     // the loader doubles spheres: exact ties are real
                            const uint32_t ix = index[sl + k];
                            const uint32_t have = w_idx == kWalkIndexOpen ? index[w_slot] : w_idx;
                            w_idx = have;
                            if (ix < have) {
                                w_slot = sl + (uint32_t)k;
                                w_idx = ix;
                            }
                        }
                    }
                }
            }
            }
            sp -= 1;
            cur = sp >= 0 ? (uint32_t)my_stack[sp * stack_stride] : kWalkDone;
            sp = sp < 0 ? 0 : sp;
        }
    }
}


// The same walk with three things taken out of its inner loops (A/B, round 4): a shadow walk's pruning by scene index is one
// compare against a per-ray bound (the written form made the compiler branch around a fourth LDS read of the pair for it); the
// scene indices of a leaf's blockers are read after its eight tests (one wait, not one per blocker); an exact tie in distance
// leaves the straight line through a branch the whole wavefront skips when nobody has one.
RT_DEV void walk_pairs_b(const float4 *s_pairs, const float4 *s_slots, const uint32_t *index, uint16_t *my_stack, int stack_stride,
                         uint32_t n_always, V3 o, V3 d, const BvhRay &R, bool shadow, int budget, int round_len, uint32_t &cur,
                         int &sp, float &w_far, uint32_t &w_idx, uint32_t &w_slot) {
    while (cur != kWalkDone && budget > 0) {
        for (int round = round_len; cur < kBvhLeafRef && budget > 0 && round > 0; --round) {
            budget -= 1;
            const float4 *pp = s_pairs + 4u * cur;
            const float4 A0 = pp[0], B0 = pp[1], A1 = pp[2], B1 = pp[3];
            float tn0, tn1;
            const bool out0 = bvh_misses(R, A0, B0, w_far, tn0), out1 = bvh_misses(R, A1, B1, w_far, tn1);
            const uint32_t prune = shadow ? w_idx : 0xffffffffu;
            const bool m0 = (int)out0 | (int)(__float_as_uint(B0.w) > prune), m1 = (int)out1 | (int)(__float_as_uint(B1.w) > prune);
            const uint32_t r0 = __float_as_uint(A0.w), r1 = __float_as_uint(A1.w);
            const bool both = !m0 & !m1, none = m0 & m1;
            const bool second_first = both ? (tn1 < tn0) : m0;
            const uint32_t near = second_first ? r1 : r0, far = second_first ? r0 : r1;
            if (both) {
                my_stack[sp * stack_stride] = (uint16_t)far;
                sp += 1;
            }
            if (none) {
                sp -= 1;
                cur = sp >= 0 ? (uint32_t)my_stack[sp * stack_stride] : kWalkDone;
                sp = sp < 0 ? 0 : sp;
            } else {
                cur = near;
            }
        }
        if (cur != kWalkDone && cur >= kBvhLeafRef) {
            const uint32_t sl = n_always + (uint32_t)kBvhLeaf * (cur & (kBvhLeafRef - 1u));
            uint32_t blockers = 0u;
#pragma unroll
            for (int half = 0; half < kBvhLeaf; half += 4) {
                HitPre p[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) p[k4] = hit_pre(s_slots[sl + (uint32_t)(half + k4)], o, d);
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int k = half + k4;
                    if (p[k4].det >= 0.f) {
                        const HitRoots hr = hit_roots(p[k4]);
                        const bool nearer = hr.hit & (hr.t < w_far);
                        blockers |= (shadow & nearer) ? (1u << k) : 0u;
                        const bool tie = !shadow & hr.hit & (hr.t == w_far);
                        if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {
                            if (tie) {                                  // the loader doubles spheres: exact ties are real
                                const uint32_t ix = index[sl + (uint32_t)k];
                                const uint32_t have = w_idx == kWalkIndexOpen ? index[w_slot] : w_idx;
                                w_idx = have;
                                if (ix < have) {
                                    w_slot = sl + (uint32_t)k;
                                    w_idx = ix;
                                }
                            }
                        }
                        const bool take = nearer & !shadow;
                        w_far = take ? hr.t : w_far;
                        w_slot = take ? sl + (uint32_t)k : w_slot;
                        w_idx = take ? kWalkIndexOpen : w_idx;
                    }
                }
            }
            while (blockers != 0u) {                                    // shadow ray: the lowest scene index that blocks (.cl:234-247)
                const uint32_t k = (uint32_t)__builtin_ctz(blockers);
                blockers &= blockers - 1u;
                const uint32_t ix = index[sl + k];
                w_idx = ix < w_idx ? ix : w_idx;
            }
            sp -= 1;
            cur = sp >= 0 ? (uint32_t)my_stack[sp * stack_stride] : kWalkDone;
            sp = sp < 0 ? 0 : sp;
        }
    }
}


// walk_pairs_b with three more trims (A/B): no per-lane step budget (a walk runs to its end within the trip: the budget of 64
// was never reached), the kept sibling written above the top whether it is kept or not (no branch; `both` only moves the
// stack pointer), and the root half of a leaf's sphere tests behind a wave ballot instead of a per-lane branch.
RT_DEV void walk_pairs_c(const float4 *s_pairs, const float4 *s_slots, const uint32_t *index, uint16_t *my_stack, int stack_stride,
                         uint32_t n_always, V3 o, V3 d, const BvhRay &R, bool shadow, int budget, int round_len, uint32_t &cur,
                         int &sp, float &w_far, uint32_t &w_idx, uint32_t &w_slot) {
    (void)budget;
    while (cur != kWalkDone) {
        for (int round = round_len; cur < kBvhLeafRef && round > 0; --round) {
            const float4 *pp = s_pairs + 4u * cur;
            const float4 A0 = pp[0], B0 = pp[1], A1 = pp[2], B1 = pp[3];
            float tn0, tn1;
            const bool out0 = bvh_misses(R, A0, B0, w_far, tn0), out1 = bvh_misses(R, A1, B1, w_far, tn1);
            const uint32_t prune = shadow ? w_idx : 0xffffffffu;
            const bool m0 = (int)out0 | (int)(__float_as_uint(B0.w) > prune), m1 = (int)out1 | (int)(__float_as_uint(B1.w) > prune);
            const uint32_t r0 = __float_as_uint(A0.w), r1 = __float_as_uint(A1.w);
            const bool both = !m0 & !m1, none = m0 & m1;
            const bool second_first = both ? (tn1 < tn0) : m0;
            const uint32_t near = second_first ? r1 : r0, far = second_first ? r0 : r1;
            my_stack[sp * stack_stride] = (uint16_t)far;       // (dead unless `both`: the entry above the top)
            sp += both ? 1 : 0;
            if (none) {
                sp -= 1;
                cur = sp >= 0 ? (uint32_t)my_stack[sp * stack_stride] : kWalkDone;
                sp = sp < 0 ? 0 : sp;
            } else {
                cur = near;
            }
        }
        if (cur != kWalkDone && cur >= kBvhLeafRef) {
            const uint32_t sl = n_always + (uint32_t)kBvhLeaf * (cur & (kBvhLeafRef - 1u));
            uint32_t blockers = 0u;
#pragma unroll
            for (int half = 0; half < kBvhLeaf; half += 4) {
                HitPre p[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) p[k4] = hit_pre(s_slots[sl + (uint32_t)(half + k4)], o, d);
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int k = half + k4;
                    if (wave_any_nonneg(p[k4].det)) {
                        const HitRoots hr = hit_roots(p[k4]);
                        const bool nearer = hr.hit & (hr.t < w_far);
                        blockers |= (shadow & nearer) ? (1u << k) : 0u;
                        const bool tie = !shadow & hr.hit & (hr.t == w_far);
                        if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {
                            if (tie) {                                  // the loader doubles spheres: exact ties are real
                                const uint32_t ix = index[sl + (uint32_t)k];
                                const uint32_t have = w_idx == kWalkIndexOpen ? index[w_slot] : w_idx;
                                w_idx = have;
                                if (ix < have) {
                                    w_slot = sl + (uint32_t)k;
                                    w_idx = ix;
                                }
                            }
                        }
                        const bool take = nearer & !shadow;
                        w_far = take ? hr.t : w_far;
                        w_slot = take ? sl + (uint32_t)k : w_slot;
                        w_idx = take ? kWalkIndexOpen : w_idx;
                    }
                }
            }
            while (blockers != 0u) {                                    // shadow ray: the lowest scene index that blocks (.cl:234-247)
                const uint32_t k = (uint32_t)__builtin_ctz(blockers);
                blockers &= blockers - 1u;
                const uint32_t ix = index[sl + k];
                w_idx = ix < w_idx ? ix : w_idx;
            }
            sp -= 1;
            cur = sp >= 0 ? (uint32_t)my_stack[sp * stack_stride] : kWalkDone;
            sp = sp < 0 ? 0 : sp;
        }
    }
}



// walk_pairs_c trimmed further (A/B; needs the materials by slot: a closest-hit walk keeps no scene index).  Pair step: which
// child comes first is mask logic, not selects of materialised booleans.  Leaf step: a closest-hit ray tracks distance and slot
// only; "some sphere's distance equalled the best so far" is one accumulated mask for the whole leaf, and the rare wavefront
// that has one settles it afterwards -- the reference's loop (.cl:215-232: strictly nearer wins, scene order breaks ties) is
// the lexicographic minimum over (distance, scene index), so the spheres of this leaf at exactly the final distance are
// compared by index with the holder of the slot.
RT_DEV void walk_pairs_d(const float4 *s_pairs, const float4 *s_slots, const uint32_t *index, uint16_t *my_stack, int stack_stride,
                         uint32_t n_always, V3 o, V3 d, const BvhRay &R, bool shadow, int round_len, uint32_t &cur, int &sp, float &w_far,
                         uint32_t &w_idx, uint32_t &w_slot) {
    while (cur != kWalkDone) {
        for (int round = round_len; cur < kBvhLeafRef && round > 0; --round) {
            const float4 *pp = s_pairs + 4u * cur;
            const float4 A0 = pp[0], B0 = pp[1], A1 = pp[2], B1 = pp[3];
            float tn0, tn1;
            const bool out0 = bvh_misses(R, A0, B0, w_far, tn0), out1 = bvh_misses(R, A1, B1, w_far, tn1);
            const uint32_t prune = shadow ? w_idx : 0xffffffffu;
            const bool m0 = (int)out0 | (int)(__float_as_uint(B0.w) > prune), m1 = (int)out1 | (int)(__float_as_uint(B1.w) > prune);
            const bool both = !m0 & !m1, none = m0 & m1;
            const bool second_first = (int)m0 | ((int)!m1 & (int)(tn1 < tn0));      // the second child first: the first is missed, or both are hit and it is nearer
            const uint32_t r0 = __float_as_uint(A0.w), r1 = __float_as_uint(A1.w);
            const uint32_t near = second_first ? r1 : r0, far = second_first ? r0 : r1;
            my_stack[sp * stack_stride] = (uint16_t)far;       // (dead unless `both`: the entry above the top)
            sp += both ? 1 : 0;
            if (none) {
                sp -= 1;
                cur = sp >= 0 ? (uint32_t)my_stack[sp * stack_stride] : kWalkDone;
                sp = sp < 0 ? 0 : sp;
            } else {
                cur = near;
            }
        }
        if (cur != kWalkDone && cur >= kBvhLeafRef) {
            const uint32_t sl = n_always + (uint32_t)kBvhLeaf * (cur & (kBvhLeafRef - 1u));
            uint32_t blockers = 0u;
            bool tie = false;
#pragma unroll
            for (int half = 0; half < kBvhLeaf; half += 4) {
                HitPre p[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) p[k4] = hit_pre(s_slots[sl + (uint32_t)(half + k4)], o, d);
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int k = half + k4;
                    if (wave_any_nonneg(p[k4].det)) {
                        const HitRoots hr = hit_roots(p[k4]);
                        const bool nearer = hr.hit & (hr.t < w_far);
                        blockers |= nearer ? (1u << k) : 0u;                    // (a shadow ray's blockers; a closest-hit ray ignores the mask)
                        tie = (int)tie | ((int)hr.hit & (int)(hr.t == w_far));
                        const bool take = nearer & !shadow;
                        w_far = take ? hr.t : w_far;
                        w_slot = take ? sl + (uint32_t)k : w_slot;
                    }
                }
            }
            if (shadow) {
                while (blockers != 0u) {                                // the lowest scene index that blocks (.cl:234-247)
                    const uint32_t k = (uint32_t)__builtin_ctz(blockers);
                    blockers &= blockers - 1u;
                    const uint32_t ix = index[sl + k];
                    w_idx = ix < w_idx ? ix : w_idx;
                }
            }
            if (__builtin_amdgcn_ballot_w64(tie & !shadow) != 0ull) {
                if (tie & !shadow) {                                    // the loader doubles spheres: exact ties are real
                    uint32_t have = index[w_slot];
                    for (int k = 0; k < kBvhLeaf; ++k) {
                        const HitPre pk = hit_pre(s_slots[sl + (uint32_t)k], o, d);
                        if (pk.det >= 0.f) {
                            const HitRoots hk = hit_roots(pk);
                            if (hk.hit && hk.t == w_far) {
                                const uint32_t ix = index[sl + (uint32_t)k];
                                if (ix < have) {
                                    have = ix;
                                    w_slot = sl + (uint32_t)k;
                                }
                            }
                        }
                    }
                }
            }
            sp -= 1;
            cur = sp >= 0 ? (uint32_t)my_stack[sp * stack_stride] : kWalkDone;
            sp = sp < 0 ? 0 : sp;
        }
    }
}

#if RT_OPT_WALK >= 3
#include "rt_walk2.inc.h"      // the second form of the kernel (its own kernel body); RT_OPT_WALK 4: with its census
#else
#undef RT_WALK_COUNT
#undef RT_WALK_CLOCK

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
    constexpr uint32_t kNone = kWalkDone;
#if RT_OPT_GLOBAL_TABLES
    // tables too large for LDS: pairs, slots and lights are read where they lie; staged: hdr | one stack per lane
    const float4 *s_pairs = P.bvh.blob + bvh_pairs_at(n_slots);
    const float4 *s_slots = P.bvh.blob + bvh_slots_at();
    uint16_t *s_stack = reinterpret_cast<uint16_t *>(s_hdr + 2);
    const float4 *s_lightA = P.scene.lightA, *s_lightB = P.scene.lightB;
    float4 *s_emis = s_hdr + 2 + stack_f4;        // (never read: the host keeps mat_in_lds off)
    float4 *s_colr = s_emis;
#else
    // staged: hdr | pairs | slots | one stack of P.bvh.stack_depth u16 per lane ([level][lane])
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
    // the lane's pixel: its wavefront's 8x8 square, or the pixel of this workgroup's rank in the order by cost of the region
    // (32 x P.deal_rows pixels) its tile lies in (P.deal; rt_trace.inc.h)
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
    unsigned long long c_tests = 0;

    // ---- lane state ---------------------------------------------------------------------------
    enum { kNew = 0, kClosest = 1, kShadow = 2, kLights = 3 };
    int st = kNew;
    V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f);     // the ray in flight: the path's, or the shadow ray (o = hit point)
    V3 thr = mk(1.f, 1.f, 1.f), rad = mk(0.f, 0.f, 0.f);
    int depth = 0;
    bool after_specular = true;
    // the walk: next node, distance bound, best sphere so far (scene index and slot)
    uint32_t cur = kNone;               // what the lane looks at next: a pair, kBvhLeafRef | leaf, or nothing (walk over)
    int sp = 0;                         // entries on its stack
    uint32_t w_idx = 0xffffffffu, w_slot = 0;
    float w_far = 0.f;
    BvhRay R = bvh_ray(s_hdr, o, d);
    // a diffuse hit being lit: its normal, the light sum, the light in flight and what it adds if unblocked
    V3 nl = mk(0.f, 0.f, 1.f), ld = mk(0.f, 0.f, 0.f);
    uint32_t lj = 0;
    float l_k = 0.f;
    unsigned long long unused_roots = 0;
#if RT_OPT_WALK == 2
    // census instance: wave-level trips and lane participation of the two phases, and where the clock goes
    // cen[0/1] pair steps (two box tests each) per wavefront / per lane, [2/3] leaf steps (kBvhLeaf sphere tests each),
    // [4/5] shade phases, [6/7] clock ticks in the walk / in shading, [8] loop trips, [9] sphere tests of the always-list sweeps
    unsigned long long cen[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#define RT_WALK_COUNT(k)                                                                         \
    do {                                                                                         \
        const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true);                       \
        if (lane == __ffsll((long long)act_) - 1) cen[k] += 1ull;                                \
        cen[(k) + 1] += 1ull;                                                                    \
    } while (0)
#define RT_WALK_CLOCK(k, t0)                                                                     \
    do {                                                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                            \
        const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true);                       \
        if (lane == __ffsll((long long)act_) - 1) cen[k] += now_ - (t0);                         \
    } while (0)
#else
#define RT_WALK_COUNT(k)
#define RT_WALK_CLOCK(k, t0)
#endif

    for (;;) {
        if (st == kNew && s >= s_end) break;
#if RT_OPT_WALK == 2
        cen[8] += (lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) ? 1ull : 0ull;
        const unsigned long long t_trip = __builtin_amdgcn_s_memtime();
#endif

        // ---- T: walk ----
        // (walk_pairs above: up to P.walk_steps pair steps of this lane's walk, leaf steps in between)
        if (cur != kNone) {
#if RT_OPT_WALK == 2
            unsigned long long *cen_p = cen;
#else
            unsigned long long *cen_p = nullptr;
#endif
            walk_pairs(s_pairs, s_slots, s_index, s_stack + tid, kBlockThreads, n_always, o, d, R, st == kShadow, P.walk_steps,
                       P.walk_round, cur, sp, w_far, w_idx, w_slot, cen_p);
        }

#if RT_OPT_WALK == 2
        RT_WALK_CLOCK(6, t_trip);
        const unsigned long long t_s = __builtin_amdgcn_s_memtime();
#endif
        // ---- S: lanes whose walk has ended, once enough of them wait ----
        const bool ready = cur == kNone;
        const unsigned long long br = __builtin_amdgcn_ballot_w64(ready);
        const unsigned long long bw = __builtin_amdgcn_ballot_w64(!ready);
        const bool go = (__popcll(br) >= P.regen_gate) || (bw == 0ull);
        if (ready && go) {
            RT_WALK_COUNT(4);
            bool path_done = false;
            bool start_closest = false;
            if (st == kShadow) {
                // ---- the shadow ray of light lj - 1 has its answer, .cl:297-301 ----
                const bool blocked = w_idx < n;
                c_tests += blocked ? w_idx + 1u : n;
                if (!blocked) {
                    const float4 lb = s_lightB[lj - 1u];
                    ld = add(ld, scale(mk(lb.x, lb.y, lb.z), l_k));
                }
                st = kLights;
            } else if (st == kClosest) {
                c_closest += 1;
                if (!(w_far < 1e20f)) {
                    path_done = true;                                              // miss, .cl:327-330
                } else {
                    const float4 ge = s_slots[w_slot];
                    const uint32_t id = w_idx == kWalkIndexOpen ? s_index[w_slot] : w_idx;
                    float4 em4, co4;
                    if (P.mat_in_lds) {
                        em4 = s_emis[id];
                        co4 = s_colr[id];
                        asm volatile("; materials from LDS" : "+v"(em4.x));
                    } else {
                        em4 = P.scene.emis[id];
                        co4 = P.scene.colr[id];
                    }
                    const V3 em = mk(em4.x, em4.y, em4.z);
                    const V3 col = mk(co4.x, co4.y, co4.z);
                    const int refl = __float_as_int(em4.w);
                    const V3 hp = add(o, scale(d, w_far));                         // .cl:338-340
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
                        else start_closest = true;
                    }
                }
            }
            // ---- next-event estimation, .cl:249-303: the lights one by one, each with its two draws ----
            while (st == kLights) {
                if (lj == n_lights) {
                    rad = add(rad, mul(thr, ld));                                  // .cl:377-378
                    // cosine-weighted bounce, .cl:383-411
                    float u = next_random(s0, s1);
                    float r2 = next_random(s0, s1);
                    c_draws += 2;
                    float r2s = rt_sqrt_unit(r2);
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
                    nd = add(nd, scale(w, rt_sqrt_unit(1 - r2)));
                    d = nd;
                    depth += 1;
                    st = kNew;                                                     // (leaves the light loop)
                    if (depth >= kMaxDepth) path_done = true;
                    else start_closest = true;
                    break;
                }
                const float4 lb = s_lightB[lj];
                V3 sd;
                float len, numer;
                const bool want = sample_light(s_lightA[lj], lb, s0, s1, c_draws, o, nl, sd, len, numer);
                lj += 1u;
                if (want) {
                    // ---- shadow ray, any hit, .cl:234-247: the large spheres now, the tree in the trips to come ----
                    c_shadow += 1;
                    l_k = rt_div(numer, len * len);                                // .cl:297 (used only if nothing blocks)
                    d = sd;
                    w_far = len - RT_EPS;
                    const uint32_t first_large = sweep_any(s_slots, n_always, o, d, w_far, unused_roots);
                    w_idx = first_large < n_always ? s_index[first_large] : n;
#if RT_OPT_WALK == 2
                    cen[9] += first_large < n_always ? first_large + 1u : n_always;
#endif
                    R = bvh_ray(s_hdr, o, d);
                    cur = root_ref;
                    sp = 0;
                    st = kShadow;
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
                start_closest = false;
            }
            if (st == kNew && !start_closest && s < s_end) {
                // ---- camera ray, .cl:494-549.  The camera (12 floats) and 1/w, 1/h come from LDS here, once per sample
                // (broadcast reads), instead of occupying 14 scalar registers through the loop: the loop overfills the scalar
                // file and its spills go to lanes of a vector register the allocator then lacks ----
                const float4 c0 = s_cam[0], c1 = s_cam[1], c2 = s_cam[2], c3 = s_cam[3];
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
                start_closest = true;
            }
            if (start_closest) {
                // ---- closest hit, .cl:215-232: the large spheres now, the tree in the trips to come ----
                float t = 1e20f;
                uint32_t slot = 0;
                sweep_closest(s_slots, n_always, o, d, t, slot, unused_roots);
#if RT_OPT_WALK == 2
                cen[9] += n_always;
#endif
                w_far = t;
                w_slot = slot;
                w_idx = (t < 1e20f) ? kWalkIndexOpen : 0xffffffffu;       // (the always-list winner's index is read if it stays the winner)
                R = bvh_ray(s_hdr, o, d);
                cur = root_ref;
                sp = 0;
                st = kClosest;
            }
        }
#if RT_OPT_WALK == 2
        RT_WALK_CLOCK(7, t_s);
#endif
    }
#if RT_OPT_WALK == 2
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
    unsigned long long tests64 = c_tests + (unsigned long long)c_closest * n;
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
#endif   // RT_OPT_WALK < 3

#if RT_OPT_WALK == 1 && RT_DIAGNOSTICS && defined(RT_WALK_RAYS_KERNEL_NAME)
// Diagnostics (rt_debug_walk_rays): arbitrary rays through the walk AND through the plain sweep over the full table,
// one lane per ray -- the unit test of the one-sided culling with rays a path tracer produces once in 10^8 (exact
// zeros and denormals in the direction, origins on box planes, far away, non-unit and non-finite directions).
//   rays[2i] = { o.xyz, t_max }, rays[2i+1] = { d.xyz, bits(shadow) }
//   out[i]   = closest: { bits(t) or ~0, scene index or ~0 } of the walk, then of the sweep;
//              shadow:  { first blocking index or n, 0 } of the walk, then of the sweep
extern "C" __global__ void __launch_bounds__(256) RT_WALK_RAYS_KERNEL_NAME(const LaunchParams P, const float4 *rays, uint32_t n_rays,
                                                                            uint4 *out) {
    constexpr int kThreads = 256;
    extern __shared__ float4 lds[];
    const uint32_t n = P.scene.n_spheres, n_always = P.bvh.n_always, n_slots = P.bvh.n_slots;
    const uint32_t *g_index = reinterpret_cast<const uint32_t *>(P.bvh.blob + bvh_index_at(n_slots));
    float4 *s_hdr = lds;
    const uint32_t n_pairs = P.bvh.n_leaves - 1u;
    float4 *s_pairs = s_hdr + 2;
    float4 *s_slots = s_pairs + 4 * n_pairs;
    uint16_t *s_stack = reinterpret_cast<uint16_t *>(s_slots + n_slots);
    const uint32_t root_ref = n_pairs ? P.bvh.root : kBvhLeafRef;
    const int tid = threadIdx.x;
    if (tid < 2) s_hdr[tid] = P.bvh.blob[tid];
    {
        const float4 *g_pairs = P.bvh.blob + bvh_pairs_at(n_slots);
        const float4 *g_slots = P.bvh.blob + bvh_slots_at();
        for (uint32_t i = tid; i < 4u * n_pairs; i += kThreads) s_pairs[i] = g_pairs[i];
        for (uint32_t i = tid; i < n_slots; i += kThreads) s_slots[i] = g_slots[i];
    }
    __syncthreads();
    const uint32_t rounds = (n_rays + gridDim.x * kThreads - 1) / (gridDim.x * kThreads);
    for (uint32_t k = 0; k < rounds; ++k) {                 // (every lane takes part in every round: the sweeps use wave ballots)
        const uint32_t i = (k * gridDim.x + blockIdx.x) * kThreads + (uint32_t)tid;
        const bool live = i < n_rays;
        const float4 ra = live ? rays[2 * i] : make_float4(0.f, 0.f, 0.f, 1.f);
        const float4 rb = live ? rays[2 * i + 1] : make_float4(0.f, 0.f, 1.f, 0.f);
        const V3 o = mk(ra.x, ra.y, ra.z), d = mk(rb.x, rb.y, rb.z);
        const bool shadow = __float_as_uint(rb.w) != 0u;
        unsigned long long roots = 0;
        uint4 res;
        const BvhRay R = bvh_ray(s_hdr, o, d);
        uint32_t cur = root_ref, w_slot = 0, w_idx;
        int sp = 0;
        float w_far;
        // the two kinds of ray diverge here; each sweep's ballots see the lanes of its own kind
        if (shadow) {
            const uint32_t first_large = sweep_any(s_slots, n_always, o, d, ra.w, roots);
            w_idx = first_large < n_always ? g_index[first_large] : n;
            w_far = ra.w;
            walk_pairs(s_pairs, s_slots, g_index, s_stack + tid, kThreads, n_always, o, d, R, true, 0x7fffffff, 3, cur, sp, w_far, w_idx,
                       w_slot, nullptr);
            const uint32_t ref = sweep_any(P.scene.geom, n, o, d, ra.w, roots);
            res = make_uint4(w_idx, 0u, ref, 0u);
        } else {
            float t = 1e20f;
            uint32_t slot = 0;
            sweep_closest(s_slots, n_always, o, d, t, slot, roots);
            w_far = t;
            w_slot = slot;
            w_idx = (t < 1e20f) ? kWalkIndexOpen : 0xffffffffu;
            walk_pairs(s_pairs, s_slots, g_index, s_stack + tid, kThreads, n_always, o, d, R, false, 0x7fffffff, 3, cur, sp, w_far, w_idx,
                       w_slot, nullptr);
            if (w_idx == kWalkIndexOpen) w_idx = g_index[w_slot];
            float t_ref = 1e20f;
            uint32_t id_ref = 0;
            sweep_closest(P.scene.geom, n, o, d, t_ref, id_ref, roots);
            const bool hit = w_far < 1e20f, hit_ref = t_ref < 1e20f;
            res = make_uint4(hit ? __float_as_uint(w_far) : 0xffffffffu, hit ? w_idx : 0xffffffffu,
                             hit_ref ? __float_as_uint(t_ref) : 0xffffffffu, hit_ref ? id_ref : 0xffffffffu);
        }
        if (live) out[i] = res;
    }
}
#endif
