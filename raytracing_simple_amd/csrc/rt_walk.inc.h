// rt_walk.inc.h -- the path-trace kernel for large scenes (RT_OPT_WALK), included by rt_trace.inc.h in place of its own
// kernel body; everything above it there (vector helpers, RNG, the sphere test, the sweeps) is shared.
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
//   T  every lane with a walk in flight walks it to its end (walk_pairs below; closest-hit and shadow rays run the same
//      loop, they differ in how a candidate updates the state);
//   S  lanes whose walk has ended (or that have no ray) do what comes next for them -- process the hit, sample the next
//      light or bounce, finish the sample, start a camera ray -- once P.regen_gate of them are waiting or nobody is
//      walking, and then join T again.  Each piece of work runs ONCE per phase for whoever needs it: the two random draws,
//      the sine / cosine and the square root that a light sample and a diffuse bounce both begin with are one section for
//      both (a wavefront's lanes are typically half back from a closest-hit walk -- light sample next -- and half back from
//      a shadow walk -- bounce next), and every new ray, closest-hit or shadow, starts in one place (always-list sweep, ray
//      set-up).
//
// Nothing here depends on which trip a lane does what: per pixel the sequence of operations and random draws is the
// reference's.  RT_OPT_WALK 2 is the census instance of the diagnostics build (steps executed, lanes taking part, clock
// shares: counters[20..29], and how many lanes the steps run with: counters[8..15]); with RT_OPT_GLOBAL_TABLES the pairs
// and slots are read where they lie in HBM / L2.  Round 4 measured this kernel's time to follow the number of instructions
// a wavefront issues, of every kind: the walk's inner loops are written to keep branches, scalar
// mask bookkeeping and waits out of them.  Other forms (depth-first nodes with skip links; a walk loop under wave-uniform
// control with branch-free steps; leaf steps done cooperatively by eight lanes per ray; ending a trip's walk phase early)
// were measured and dropped: DESIGN.md section 5 names the commits.

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
// the pair the walk starts at (kBvhLeafRef: the tree is one leaf): the low half of the header's last word
RT_DEV uint32_t bvh_root(const float4 *s_hdr) { return __float_as_uint(s_hdr[1].w) & 0xffffu; }
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
#if RT_OPT_PACKED_PAIRS
    // The packed pair table stores a box plane as a grid coordinate q: plane = r0 + q * scale (the frame: s_hdr[2], s_hdr[3]).  Its slab distance
    // (plane - (o +- pad)) / d = q * (scale / d) + (r0 - (o +- pad)) / d is ONE fused multiply-add on q with the two constants folded per ray
    // into the places of 1 / d and -(o +- pad) / d -- bvh_misses_upto then runs unchanged on the grid coordinates.  Rounding: the plane the
    // computed distance belongs to lies within u (4 far + 6 |o|) of the stored one (one rounding each of scale / d, of the folded offset -- a
    // fused multiply-add on r0 --, and of the step's own), inside the pad's linear term 64 u (|o| + far + r_max) like the unfolded form's
    // 4u |plane - o| + 2u |o|.  A ray that is not `sane` (infinite pad: it visits everything) gets 0 and -+infinity: q * 0 -+ inf leaves every
    // axis unconstrained whatever the frame holds (scale / d may overflow there).
    {
        const float4 f0 = s_hdr[2], f1 = s_hdr[3];
        const V3 k = mk(f1.x * R.inv.x, f1.y * R.inv.y, f1.z * R.inv.z);
        const V3 cl = mk(__builtin_fmaf(f0.x, R.inv.x, R.clo.x), __builtin_fmaf(f0.y, R.inv.y, R.clo.y), __builtin_fmaf(f0.z, R.inv.z, R.clo.z));
        const V3 ch = mk(__builtin_fmaf(f0.x, R.inv.x, R.chi.x), __builtin_fmaf(f0.y, R.inv.y, R.chi.y), __builtin_fmaf(f0.z, R.inv.z, R.chi.z));
        R.inv = sane ? k : mk(0.f, 0.f, 0.f);
        R.clo = sane ? cl : mk(-inf, -inf, -inf);
        R.chi = sane ? ch : mk(inf, inf, inf);
    }
#endif
    return R;
}
// True when the ray stretch [-tback, t_far + tback] misses the grown box for certain.  A slab distance is
// fma(plane, 1/d, -(o +- pad)/d): the plane this computed distance really belongs to -- o +- pad + t d, exactly -- lies
// within  4u |plane - o| + 2u |o|  of the box's (one ulp of v_rcp_f32, the rounded shifted origin, its rounded product,
// the fused operation's own rounding), which the pad's linear term covers with the rest (above); so the point X of
// the derivation above, which is inside the grown box by that margin, is between the computed planes on every axis and
// its parameter inside [tn, tf].  Minimum and maximum drop NaN operands (a direction component of 0 against a plane
// through the origin): that axis then does not constrain.
// (minimum and maximum as the instructions themselves -- v_min_f32 / v_max_f32 / v_min3 / v_max3 return the operand that is a
// number when one is not -- without the canonicalising copies the compiler puts in front of fminf / fmaxf)
RT_DEV float lean_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
RT_DEV float lean_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
RT_DEV float lean_max_neg(float a, float b) {          // max(a, -b)
    float r;
    asm("v_max_f32 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
RT_DEV float lean_min3(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
RT_DEV float lean_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// `t_hi` = t_far + R.tback: the far end of the stretch of the ray the box has to meet (formed once per round of steps, not per box)
RT_DEV bool bvh_misses_upto(const BvhRay &R, float4 A, float4 B, float t_hi, float &t_near) {
    const float x0 = __builtin_fmaf(A.x, R.inv.x, R.clo.x), x1 = __builtin_fmaf(B.x, R.inv.x, R.chi.x);
    const float y0 = __builtin_fmaf(A.y, R.inv.y, R.clo.y), y1 = __builtin_fmaf(B.y, R.inv.y, R.chi.y);
    const float z0 = __builtin_fmaf(A.z, R.inv.z, R.clo.z), z1 = __builtin_fmaf(B.z, R.inv.z, R.chi.z);
    const float tn = lean_max3(lean_min(x0, x1), lean_min(y0, y1), lean_max_neg(lean_min(z0, z1), R.tback));
    const float tf = lean_min3(lean_max(x0, x1), lean_max(y0, y1), lean_min(lean_max(z0, z1), t_hi));
    t_near = tn;                        // where the ray enters the grown box (an ordering hint, nothing more)
    return tn > tf;
}
RT_DEV bool bvh_misses(const BvhRay &R, float4 A, float4 B, float t_far, float &t_near) {
    return bvh_misses_upto(R, A, B, t_far + R.tback, t_near);
}
RT_DEV bool bvh_misses(const BvhRay &R, float4 A, float4 B, float t_far) {
    float unused;
    return bvh_misses(R, A, B, t_far, unused);
}

// ---- the walk over sibling pairs (rt_device.h BvhTables `pairs`), nearer child first ----
// One pair step loads a pair (64 bytes), tests both boxes against the ray's stretch, goes on with the nearer of the
// children that are hit and keeps the other on the lane's stack (16 bits per entry, [level][lane]); with neither hit
// it takes the last kept one.  Which child comes first only decides how soon the bound shrinks: every sphere whose
// chain of boxes the ray meets is still tested, so the result is the same set of candidates run through the same rule.
// Shadow rays look for the LOWEST blocking scene index (that is what .cl:234-247 returns at): a subtree that only
// holds higher indices than the best so far is skipped -- one compare against a per-ray bound.  At most `round_len` pair
// steps are taken in a row before the leaf step of the lanes that hold a leaf -- a lane that is still looking goes on
// looking in the next round instead of keeping the others waiting (the number of steps to the next leaf has a long tail).
// A leaf step runs the leaf's eight spheres through the reference's test, in two halves of four (all eight discriminants
// at once are registers the kernel does not have), the root half behind a wave ballot.  What needs memory or is rare
// leaves the loop over the spheres: the scene indices of a shadow ray's blockers are read after the tests (one wait, not
// one per blocker), and an exact tie in distance (the .scn loader doubles spheres: real) takes a branch the whole
// wavefront skips when nobody has one.  The kept sibling is written above the stack's top whether it is kept or not (no
// branch; `both` only moves the stack pointer: the entry above the top is dead).
// cen (census instance only): [0] pair steps of the wavefront, [1] of this lane, [2]/[3] leaf steps; hist: steps by lanes.
constexpr uint32_t kWalkDone = 0xffffffffu;
// `tail` (0 = off): the walk phase ends for everybody once no more than `tail` lanes are still walking (a wave ballot after each
// round); their walks are lane state and go on in the next trip, beside the walks the other lanes start in between.
// `plane` (RT_OPT_PAIR_PLANES): the staged pairs lie in four planes of `plane` records (part k of pair m at [k * plane + m]), so
// that the lanes of a ds_read_b128 group, which sit at different pairs, spread over sixteen 16-byte slots of a bank row, not four.
RT_DEV void walk_pairs(const float4 *s_pairs, const float4 *s_slots, const uint32_t *index, uint16_t *my_stack, int stack_stride,
                       uint32_t n_always, V3 o, V3 d, const BvhRay &R, bool shadow, int round_len, uint32_t &cur, int &sp, float &w_far,
                       uint32_t &w_idx, uint32_t &w_slot, unsigned long long *cen, unsigned long long *hist, int tail = 0, uint32_t plane = 0,
                       bool once = false, const float4 *s_top = nullptr, uint32_t n_top = 0u) {
    const int lane_ = threadIdx.x & 63;
    uint32_t kind_m = shadow ? 0xffffffffu : 0u;        // all ones: a shadow ray (as a value the compiler does not see through)
    asm("" : "+v"(kind_m));
#if RT_OPT_PREFETCH
    // A/B ("a prefetch of both children's pairs one level ahead"): as soon as a pair's two references are in registers -- before
    // its box tests -- one dword of EACH child's record is requested (the child's pair, or its leaf's first sphere): the line is then on its way
    // into the vector L1 / L2 while the boxes are tested, whichever child the walk takes next and for the sibling it keeps.  The values are
    // never used: they are held in two registers until the next step has issued its own loads (loads return in order, so that costs no wait).
    float pf0 = 0.f, pf1 = 0.f;
#endif
    while (cur != kWalkDone) {
        for (int round = round_len; cur < kBvhLeafRef && round > 0; --round) {
            if (cen) {
                const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true);
                const int n_ = __popcll(act_);
                if (lane_ == __ffsll((long long)act_) - 1) {
                    cen[0] += 1ull;
                    hist[4 + (n_ <= 8 ? 0 : (n_ <= 16 ? 1 : (n_ <= 32 ? 2 : 3)))] += 1ull;
                }
                cen[1] += 1ull;
            }
#if RT_OPT_PACKED_PAIRS
            // 32 bytes per pair instead of 64 (rt_device.h BvhTables::packed_at): the kernel is bound by the bytes its lanes pull through the
            // vector memory path -- 64 bytes per clock and CU, whatever hits where -- and has vector ALU to spare, so
            // the boxes travel as 16-bit grid coordinates and are tested as such (bvh_ray folds the frame into the ray's constants)
            float4 A0, B0, A1, B1;
            {
                const uint4 *pq = reinterpret_cast<const uint4 *>(s_pairs) + 2u * cur;
                const uint4 w0 = pq[0], w1 = pq[1];
                A0 = make_float4((float)(w0.x & 0xffffu), (float)(w0.x >> 16), (float)(w0.y & 0xffffu), __uint_as_float(w1.z & 0xffffu));
                B0 = make_float4((float)(w0.y >> 16), (float)(w0.z & 0xffffu), (float)(w0.z >> 16), __uint_as_float((w1.w & 0xffffu) << kBvhLowShift));
                A1 = make_float4((float)(w0.w & 0xffffu), (float)(w0.w >> 16), (float)(w1.x & 0xffffu), __uint_as_float(w1.z >> 16));
                B1 = make_float4((float)(w1.x >> 16), (float)(w1.y & 0xffffu), (float)(w1.y >> 16), __uint_as_float((w1.w >> 16) << kBvhLowShift));
            }
#elif RT_OPT_TOP_PAIRS
            // the promoted top of the tree (pairs [0, n_top), breadth-first: rt_bvh.hip) from LDS, everything below it where it lies in
            // HBM / L2: a ray's first steps from the root -- and the kept siblings it returns to last -- never leave the CU
            // (two blocks under the lanes' own masks -- ds_read_b128 for the lanes in the top, global_load_dwordx4 for the others.  The pointers
            // carry their address spaces in their types: left generic, the compiler folds both blocks into ONE flat load through a selected
            // pointer, and a flat load occupies the vector memory path for every lane -- the unit this kernel is bound by)
            float4 A0, B0, A1, B1;
            if (cur < n_top) {
                typedef float F4v __attribute__((ext_vector_type(4)));
                typedef const __attribute__((address_space(3))) F4v *LdsF4;
                const LdsF4 pp = (LdsF4)(s_top + 4u * cur);
                const F4v a0 = pp[0], b0 = pp[1], a1 = pp[2], b1 = pp[3];
                A0 = make_float4(a0.x, a0.y, a0.z, a0.w); B0 = make_float4(b0.x, b0.y, b0.z, b0.w);
                A1 = make_float4(a1.x, a1.y, a1.z, a1.w); B1 = make_float4(b1.x, b1.y, b1.z, b1.w);
            } else {
                typedef float F4v __attribute__((ext_vector_type(4)));
                typedef const __attribute__((address_space(1))) F4v *GlobalF4;
                const GlobalF4 pp = (GlobalF4)(s_pairs + 4u * cur);
                const F4v a0 = pp[0], b0 = pp[1], a1 = pp[2], b1 = pp[3];
                A0 = make_float4(a0.x, a0.y, a0.z, a0.w); B0 = make_float4(b0.x, b0.y, b0.z, b0.w);
                A1 = make_float4(a1.x, a1.y, a1.z, a1.w); B1 = make_float4(b1.x, b1.y, b1.z, b1.w);
            }
#elif RT_OPT_PAIR_PLANES
            const float4 A0 = s_pairs[cur], B0 = s_pairs[cur + plane], A1 = s_pairs[cur + 2u * plane], B1 = s_pairs[cur + 3u * plane];
#else
            const float4 *pp = s_pairs + 4u * cur;
            const float4 A0 = pp[0], B0 = pp[1], A1 = pp[2], B1 = pp[3];
#endif
#if RT_OPT_PREFETCH
            asm volatile("; prefetched dwords of the step before retire here" :: "v"(pf0), "v"(pf1));
            {
                const uint32_t q0 = __float_as_uint(A0.w) & 0xffffu, q1 = __float_as_uint(A1.w) & 0xffffu;
                const float *c0 = (q0 & kBvhLeafRef) ? reinterpret_cast<const float *>(s_slots + n_always + (uint32_t)kBvhLeaf * (q0 & (kBvhLeafRef - 1u)))
                                                     : reinterpret_cast<const float *>(s_pairs + 4u * q0);
                const float *c1 = (q1 & kBvhLeafRef) ? reinterpret_cast<const float *>(s_slots + n_always + (uint32_t)kBvhLeaf * (q1 & (kBvhLeafRef - 1u)))
                                                     : reinterpret_cast<const float *>(s_pairs + 4u * q1);
                pf0 = *c0;
                pf1 = *c1;
            }
#endif
            float tn0, tn1;
            const float t_hi = w_far + R.tback;
            const bool out0 = bvh_misses_upto(R, A0, B0, t_hi, tn0), out1 = bvh_misses_upto(R, A1, B1, t_hi, tn1);
            // a shadow walk skips subtrees that hold only scene indices above its lowest blocker so far; a closest-hit walk never does
            const uint32_t prune = shadow ? w_idx : 0xffffffffu;
            const bool m0 = (int)out0 | (int)(__float_as_uint(B0.w) > prune), m1 = (int)out1 | (int)(__float_as_uint(B1.w) > prune);
            const uint32_t r0 = __float_as_uint(A0.w), r1 = __float_as_uint(A1.w);
            const bool both = !m0 & !m1, none = m0 & m1;
            const bool second_first = (int)m0 | ((int)(tn1 < tn0) & (int)!m1);      // (= both ? tn1 < tn0 : m0, as mask logic)
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
            if (cen) {
                const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true);
                const int n_ = __popcll(act_);
                if (lane_ == __ffsll((long long)act_) - 1) {
                    cen[2] += 1ull;
                    hist[n_ <= 8 ? 0 : (n_ <= 16 ? 1 : (n_ <= 32 ? 2 : 3))] += 1ull;
                }
                cen[3] += 1ull;
            }
            const uint32_t sl = n_always + (uint32_t)kBvhLeaf * (cur & (kBvhLeafRef - 1u));
            // One mask of the leaf's spheres whose scene index will be needed -- the index lies in HBM / L2 and is read after the
            // eight tests, one wait for all.  Shadow ray (.cl:234-247): the spheres that block; the lowest index among them is
            // the answer.  Closest hit (.cl:215-232: strictly nearer wins, scene order breaks ties, i.e. the lexicographic
            // minimum over (distance, scene index)): the spheres at exactly the best distance so far -- a strictly nearer hit
            // takes the slot and clears the mask, so what is left at the end ties with the final best (the .scn loader doubles
            // spheres: such ties are real, and rare).
            uint32_t need = 0u;
            uint32_t leaf_cnt = 0u;         // (census: spheres of the leaf whose discriminant is non-negative for this lane)
            constexpr int kPart = kBvhLeaf % 4 == 0 ? 4 : 3;        // (the leaf in parts of four; of three for leaves of 6)
#pragma unroll
            for (int half = 0; half < kBvhLeaf; half += kPart) {
                HitPre p[kPart];
#pragma unroll
                for (int k4 = 0; k4 < kPart; ++k4) p[k4] = hit_pre(s_slots[sl + (uint32_t)(half + k4)], o, d);
#pragma unroll
                for (int k4 = 0; k4 < kPart; ++k4) {
                    const int k = half + k4;
                    if (cen) leaf_cnt += p[k4].det >= 0.f ? 1u : 0u;
                    if (wave_any_nonneg(p[k4].det)) {
                        const HitRoots hr = hit_roots(p[k4]);
                        if (cen) {
                            if (lane_ == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) hist[10] += 1ull;
                            hist[11] += hr.hit ? 1ull : 0ull;
                            hist[12] += (hr.hit & (hr.t < w_far)) ? 1ull : 0ull;
                        }
                        const bool nearer = hr.hit & (hr.t < w_far), level = hr.hit & (hr.t == w_far);
                        const bool take = nearer & !shadow;
                        need = take ? 0u : need;
                        // (a shadow ray wants its blockers, a closest-hit ray the spheres level with its best: the lane's kind picks between
                        // two bit values -- one v_bfi -- where a select between the two CONDITIONS costs the compiler five operations)
                        const uint32_t nb = nearer ? (1u << k) : 0u, lb = level ? (1u << k) : 0u;
                        need |= (nb & kind_m) | (lb & ~kind_m);
                        w_far = take ? hr.t : w_far;
                        w_slot = take ? sl + (uint32_t)k : w_slot;
                    }
                }
            }
            if (cen) {
                hist[8] += (unsigned long long)leaf_cnt;
                for (uint32_t j = 1; j <= (uint32_t)kBvhLeaf; ++j)
                    if (__builtin_amdgcn_ballot_w64(leaf_cnt >= j) != 0ull && lane_ == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) hist[9] += 1ull;
            }
            if (__builtin_amdgcn_ballot_w64(need != 0u) != 0ull) {
                if (need != 0u) {
                    uint32_t best = shadow ? w_idx : index[w_slot];
                    do {
                        const uint32_t k = (uint32_t)__builtin_ctz(need);
                        need &= need - 1u;
                        const uint32_t ix = index[sl + k];
                        if (ix < best) {
                            best = ix;
                            w_slot = shadow ? w_slot : sl + k;
                        }
                    } while (need != 0u);
                    w_idx = shadow ? best : w_idx;
                }
            }
            sp -= 1;
            cur = sp >= 0 ? (uint32_t)my_stack[sp * stack_stride] : kWalkDone;
            sp = sp < 0 ? 0 : sp;
        }
        if (once) break;        // (RT_OPT_RAYS2: one round per call, the caller's loop decides whose ray walks next; a constant everywhere else)
#if RT_DIAGNOSTICS      // (an experiment's knob, profiles/r05_walk_ab_c3.jsonl: the product kernel's loop does not carry its compare and branch)
        if (tail > 0 && __popcll(__builtin_amdgcn_ballot_w64(cur != kWalkDone)) <= tail) break;      // (wave-uniform)
#endif
    }
}

struct Walk {             // a ray's place in the hierarchy and what it has found (lane state across loop trips)
    uint32_t cur;           // what the lane looks at next: a pair, kBvhLeafRef | leaf, or kWalkDone
    int sp;                 // entries on its stack
    float far;              // closest hit: the best distance so far; shadow ray: its length (fixed)
    uint32_t idx;           // shadow ray: the lowest blocking scene index so far (a closest-hit walk keeps no index: the material is read by slot)
    uint32_t slot;          // closest hit: slot of the best
};

#undef RT_WALK_COUNT
#undef RT_WALK_CLOCK
#undef RT_WALK_HIST

// Where a lane's path stands, in ONE register (the kernel has none to spare): what it does next (2 bits), the path's depth
// (4 bits), "the last bounce was specular" (1 bit), the light being sampled (24 bits).
struct PathCtl {
    uint32_t v;
    RT_DEV uint32_t st() const { return v & 3u; }
    RT_DEV void set_st(uint32_t x) { v = (v & ~3u) | x; }
    RT_DEV uint32_t depth() const { return (v >> 2) & 15u; }
    RT_DEV void deeper() { v += 4u; }
    RT_DEV bool after_specular() const { return (v & 64u) != 0u; }
    RT_DEV void set_after_specular(bool b) { v = b ? (v | 64u) : (v & ~64u); }
    RT_DEV uint32_t light() const { return v >> 8; }
    RT_DEV void next_light() { v += 256u; }
    RT_DEV void first_light() { v &= 0xffu; }
    RT_DEV void new_path() { v = (v & 3u) | 64u; }        // depth 0, after_specular, light 0
};

extern "C" __global__ void __launch_bounds__(64 * RT_OPT_WG_WAVES, RT_OPT_MINWAVES) RT_KERNEL_NAME(const LaunchParams P) {
    constexpr int kBlockThreads = 64 * RT_OPT_WG_WAVES;
    // RT_OPT_RAYS2 (diagnostics): TWO pixels per lane -- (x, y) and (x + 8, y): a wavefront renders 16 x 8 pixels.
    // One of a lane's two rays is `at hand` (the variables below, as ever), the other parked in a second set of registers; a lane whose
    // ray at hand has ended its walk takes up the parked one's inside the walk phase, and the shade phase runs once for either set.
    constexpr int kRaysPerLane = RT_OPT_RAYS2 ? 2 : 1;
    constexpr int kTileW = 8 * RT_OPT_WG_WAVES * kRaysPerLane;
    constexpr int kStackStride = kBlockThreads * kRaysPerLane;
    extern __shared__ float4 lds[];
    const uint32_t n = P.scene.n_spheres;
    const uint32_t n_lights = P.scene.n_lights;
    const uint32_t n_always = P.bvh.n_always, n_slots = P.bvh.n_slots;
    // (the scene index of a slot is only read for a candidate that passes the test: from HBM / L2, not staged)
    const uint32_t *s_index = reinterpret_cast<const uint32_t *>(P.bvh.blob + bvh_index_at(n_slots));
    float4 *s_hdr = lds;
    const uint32_t stack_f4 = (P.bvh.stack_depth * (uint32_t)kStackStride * 2u + 15u) / 16u;
#if RT_OPT_GLOBAL_TABLES == 2
    // the PAIRS staged (64 bytes per leaf: what a walk's chain of dependent fetches goes through), the slots -- 128 bytes per leaf, one
    // contiguous read per leaf step -- where they lie in HBM / L2: scenes whose whole tables outgrow the LDS budget but whose pairs fit it
    const uint32_t n_pairs = P.bvh.n_leaves - 1u;
    float4 *s_pairs = s_hdr + 2;
    const float4 *s_slots = P.bvh.blob + bvh_slots_at();
    uint16_t *s_stack = reinterpret_cast<uint16_t *>(s_pairs + 4 * n_pairs);
    const float4 *s_lightA = P.scene.lightA, *s_lightB = P.scene.lightB;
    float4 *s_emis = s_pairs + 4 * n_pairs + stack_f4;        // (never read: the host keeps mat_in_lds off)
    float4 *s_colr = s_emis;
#elif RT_OPT_GLOBAL_TABLES
#if RT_OPT_PACKED_PAIRS
    const float4 *s_pairs = P.bvh.blob + P.bvh.packed_at + 2;      // (the packed table's records: 2 x 16 bytes per pair, behind its frame)
#else
    const float4 *s_pairs = P.bvh.blob + bvh_pairs_at(n_slots);
#endif
    const float4 *s_slots = P.bvh.blob + bvh_slots_at();
#if RT_OPT_TOP_PAIRS
    const uint32_t n_top = P.bvh.n_top;         // staged: hdr | the promoted top of the tree (64 bytes per pair) | stacks
    float4 *s_top = s_hdr + 2;
#elif RT_OPT_PACKED_PAIRS
    constexpr uint32_t n_top = 0u;              // staged: hdr | the packed table's frame | stacks
    float4 *s_top = s_hdr + 4;
#else
    constexpr uint32_t n_top = 0u;
    float4 *s_top = s_hdr + 2;
#endif
    uint16_t *s_stack = reinterpret_cast<uint16_t *>(s_top + 4 * n_top);
    const float4 *s_lightA = P.scene.lightA, *s_lightB = P.scene.lightB;
    float4 *s_emis = s_top + 4 * n_top + stack_f4;        // (never read: the host keeps mat_in_lds off)
    float4 *s_colr = s_emis;
#else
    // staged: hdr | pairs | slots | one stack of P.bvh.stack_depth u16 per lane ([level][lane]) | lights | materials
    const uint32_t n_pairs = P.bvh.n_leaves - 1u;
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
    __shared__ unsigned long long s_wg_t0;      // the workgroup's start on the device's wall clock (10 ns ticks)
    if (tid == 5) {
        s_tile_cost = 0u;
        s_wg_t0 = __builtin_amdgcn_s_memrealtime();
    }
    if (tid == 6) {
        s_cam[0] = make_float4(P.cam.orig.x, P.cam.orig.y, P.cam.orig.z, P.cam.dir.x);
        s_cam[1] = make_float4(P.cam.dir.y, P.cam.dir.z, P.cam.x.x, P.cam.x.y);
        s_cam[2] = make_float4(P.cam.x.z, P.cam.y.x, P.cam.y.y, P.cam.y.z);
        s_cam[3] = make_float4(0.f, 0.f, P.inv_w, P.inv_h);
    }
    if (tid < 2) s_hdr[tid] = P.bvh.blob[tid];
#if RT_OPT_PACKED_PAIRS
    if (tid >= 2 && tid < 4) s_hdr[tid] = P.bvh.blob[P.bvh.packed_at + tid - 2];
#endif
#if RT_OPT_GLOBAL_TABLES == 2
    {
        const float4 *g_pairs = P.bvh.blob + bvh_pairs_at(n_slots);
        for (uint32_t i = tid; i < 4u * n_pairs; i += kBlockThreads) s_pairs[i] = g_pairs[i];
    }
#endif
#if RT_OPT_GLOBAL_TABLES == 1 && RT_OPT_TOP_PAIRS
    for (uint32_t i = tid; i < 4u * n_top; i += kBlockThreads) s_top[i] = s_pairs[i];
#endif
#if !RT_OPT_GLOBAL_TABLES
    {
        const float4 *g_pairs = P.bvh.blob + bvh_pairs_at(n_slots);
        const float4 *g_slots = P.bvh.blob + bvh_slots_at();
#if RT_OPT_PAIR_PLANES
        for (uint32_t i = tid; i < 4u * n_pairs; i += kBlockThreads) s_pairs[(i & 3u) * n_pairs + (i >> 2)] = g_pairs[i];
#else
        for (uint32_t i = tid; i < 4u * n_pairs; i += kBlockThreads) s_pairs[i] = g_pairs[i];
#endif
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
    int x = tile_bx * kTileW + wave * 8 * kRaysPerLane + (lane & 7), lrow = tile_by * kTileH + (lane >> 3);
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
    enum : uint32_t { kNew = 0, kClosest = 1, kShadow = 2, kLights = 3 };
    PathCtl ctl{ kNew | 64u };                           // (nothing to do yet; after_specular as a new path has it)
    V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f);     // the ray in flight: the path's, or the shadow ray (o = hit point)
    V3 thr = mk(1.f, 1.f, 1.f), rad = mk(0.f, 0.f, 0.f);
    Walk W{ kWalkDone, 0, 0.f, 0xffffffffu, 0u };
    BvhRay R = bvh_ray(s_hdr, o, d);
    // a diffuse hit being lit: its normal, the light sum, the light in flight and what it adds if unblocked
    V3 nl = mk(0.f, 0.f, 1.f), ld = mk(0.f, 0.f, 0.f);
    float l_k = 0.f;
    uint16_t *my_stack = s_stack + tid;
#if RT_OPT_RAYS2
    // the lane's second pixel, parked: everything a ray is
    struct Parked {
        PathCtl ctl; V3 o, d, thr, rad; Walk W; BvhRay R; V3 nl, ld; float l_k; uint16_t *stack; uint32_t s0, s1; V3 acc; int s, s_end; uint32_t xy;
    };
    Parked pk{ PathCtl{ kNew | 64u }, o, d, thr, rad, W, R, nl, ld, 0.f, s_stack + tid + kBlockThreads, 0u, 0u, mk(0.f, 0.f, 0.f), P.first_sample, P.first_sample, 0u };
    {
        const int x2 = x + 8;
        const bool valid2 = (x2 < P.w) && (lrow < P.local_rows) && (y < P.h);
        pk.xy = (uint32_t)x2 | ((uint32_t)y << 16);
        pk.s_end = valid2 ? P.first_sample + P.n_samples : P.first_sample;
        if (valid2) {
            const size_t gid2 = (size_t)y * (size_t)P.w + (size_t)x2, ci2 = (size_t)(P.h - y - 1) * (size_t)P.w + (size_t)x2;
            const uint2 sd2 = *reinterpret_cast<const uint2 *>(P.seeds_in + 2 * gid2);
            pk.s0 = sd2.x;
            pk.s1 = sd2.y;
            if (P.first_sample > 0) pk.acc = mk(P.colors[3 * ci2], P.colors[3 * ci2 + 1], P.colors[3 * ci2 + 2]);
        }
    }
    int s_end_now = s_end;          // (`s_end` and `xy` are constants of the one-ray kernel: the two-ray one swaps them with the rest)
    uint32_t xy_now_ = xy;
#define RT_XY_ xy_now_
#define RT_SWAP_(a, b) do { auto t_ = (a); (a) = (b); (b) = t_; } while (0)
#define RT_SWAP_V3_(a, b) do { RT_SWAP_((a).x, (b).x); RT_SWAP_((a).y, (b).y); RT_SWAP_((a).z, (b).z); } while (0)
    auto swap_rays = [&]() {
        RT_SWAP_(ctl.v, pk.ctl.v); RT_SWAP_V3_(o, pk.o); RT_SWAP_V3_(d, pk.d); RT_SWAP_V3_(thr, pk.thr); RT_SWAP_V3_(rad, pk.rad);
        RT_SWAP_(W.cur, pk.W.cur); RT_SWAP_(W.sp, pk.W.sp); RT_SWAP_(W.far, pk.W.far); RT_SWAP_(W.idx, pk.W.idx); RT_SWAP_(W.slot, pk.W.slot);
        RT_SWAP_V3_(R.clo, pk.R.clo); RT_SWAP_V3_(R.chi, pk.R.chi); RT_SWAP_V3_(R.inv, pk.R.inv); RT_SWAP_(R.tback, pk.R.tback);
        RT_SWAP_V3_(nl, pk.nl); RT_SWAP_V3_(ld, pk.ld); RT_SWAP_(l_k, pk.l_k); RT_SWAP_(my_stack, pk.stack);
        RT_SWAP_(s0, pk.s0); RT_SWAP_(s1, pk.s1); RT_SWAP_V3_(acc, pk.acc); RT_SWAP_(s, pk.s); RT_SWAP_(s_end_now, pk.s_end); RT_SWAP_(xy_now_, pk.xy);
    };
#else
    const int s_end_now = s_end;
#define RT_XY_ xy
#endif
#if RT_OPT_PAIR_PLANES && !RT_OPT_GLOBAL_TABLES
    const uint32_t kPlane = P.bvh.n_leaves - 1u;
#else
    const uint32_t kPlane = 0u;
#endif
#if RT_OPT_WALK == 2
    // census instance.  cen[0/1] pair steps (two box tests each) per wavefront / per lane, [2/3] leaf steps (kBvhLeaf sphere tests
    // each), [4/5] shade phases, [6/7] clock ticks in the walk / in shading, [8] loop trips, [9] sphere tests of the always-list
    // sweeps -> counters[20..29]; hist[0..3] leaf steps with 1-8 / 9-16 / 17-32 / 33-64 lanes, hist[4..7] pair steps likewise -> counters[8..15]
    unsigned long long cen[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long emu[4] = { 0, 0, 0, 0 };         // (lane 0 only) more rays than lanes, emulated on the walk phases as executed: below
    unsigned long long hist[13] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };      // [8] lanes' non-negative discriminants per leaf step, [9] the largest such count per leaf step, [10] root halves executed, [11] hits, [12] nearer hits -> counters[16..19], [31]
#define RT_WALK_COUNT(k, mask)                                                                     \
    do {                                                                                         \
        if (lane == __ffsll((long long)(mask)) - 1) cen[k] += 1ull;                              \
        cen[(k) + 1] += 1ull;                                                                    \
    } while (0)
#define RT_WALK_HIST(base, mask)                                                                   \
    do {                                                                                         \
        const int n_ = __popcll(mask);                                                           \
        if (lane == __ffsll((long long)(mask)) - 1) hist[(base) + (n_ <= 8 ? 0 : (n_ <= 16 ? 1 : (n_ <= 32 ? 2 : 3)))] += 1ull; \
    } while (0)
#define RT_WALK_CLOCK(arr, k, t0)                                                                  \
    do {                                                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                            \
        const unsigned long long act_ = __builtin_amdgcn_ballot_w64(true);                       \
        if (lane == __ffsll((long long)act_) - 1) arr[k] += now_ - (t0);                         \
    } while (0)
#else
#define RT_WALK_COUNT(k, mask)
#define RT_WALK_HIST(base, mask)
#define RT_WALK_CLOCK(arr, k, t0)
#endif

    for (;;) {
        // (a lane that has rendered its samples stays in the loop, idle, until its wavefront has: the loop's exit is a scalar branch)
#if RT_OPT_RAYS2
        if (__builtin_amdgcn_ballot_w64(!(ctl.st() == kNew && s >= s_end_now) || !(pk.ctl.st() == kNew && pk.s >= pk.s_end)) == 0ull) break;
#else
        const bool finished = ctl.st() == kNew && s >= s_end;
        if (__builtin_amdgcn_ballot_w64(!finished) == 0ull) break;
#endif
#if RT_OPT_WALK == 2
        cen[8] += (lane == __ffsll((long long)__builtin_amdgcn_ballot_w64(true)) - 1) ? 1ull : 0ull;
        const unsigned long long t_trip = __builtin_amdgcn_s_memtime();
#endif

#if RT_OPT_RAYS2
        // ---- T, two rays per lane: round by round; a lane whose ray at hand has ended its walk takes up the parked one's ----
        for (;;) {
            const bool take = (W.cur == kWalkDone) & (pk.W.cur != kWalkDone);
            if (__builtin_amdgcn_ballot_w64((W.cur != kWalkDone) | take) == 0ull) break;
            if (take) swap_rays();
            if (W.cur != kWalkDone) {
#if RT_OPT_WALK == 2
                walk_pairs(s_pairs, s_slots, s_index, my_stack, kStackStride, n_always, o, d, R, ctl.st() == kShadow, P.walk_round & 0xff, W.cur, W.sp, W.far, W.idx,
                           W.slot, cen, hist, 0, kPlane, true);
#else
                walk_pairs(s_pairs, s_slots, s_index, my_stack, kStackStride, n_always, o, d, R, ctl.st() == kShadow, P.walk_round & 0xff, W.cur, W.sp, W.far, W.idx,
                           W.slot, nullptr, nullptr, 0, kPlane, true);
#endif
            }
        }
#else
        // ---- T: every walk in flight runs to its end ----
#if RT_OPT_WALK == 2
        const unsigned long long emu_p0 = cen[1], emu_l0 = cen[3];
#endif
        if (W.cur != kWalkDone) {
#if RT_OPT_WALK == 2
            walk_pairs(s_pairs, s_slots, s_index, my_stack, kBlockThreads, n_always, o, d, R, ctl.st() == kShadow, P.walk_round & 0xff, W.cur, W.sp, W.far, W.idx,
                       W.slot, cen, hist, P.walk_round >> 8, kPlane);
#else
#if RT_OPT_TOP_PAIRS
            walk_pairs(s_pairs, s_slots, s_index, my_stack, kBlockThreads, n_always, o, d, R, ctl.st() == kShadow, P.walk_round & 0xff, W.cur, W.sp, W.far, W.idx,
                       W.slot, nullptr, nullptr, P.walk_round >> 8, kPlane, false, s_top, n_top);
#else
            walk_pairs(s_pairs, s_slots, s_index, my_stack, kBlockThreads, n_always, o, d, R, ctl.st() == kShadow, P.walk_round & 0xff, W.cur, W.sp, W.far, W.idx,
                       W.slot, nullptr, nullptr, P.walk_round >> 8, kPlane);
#endif
#endif
        }

#if RT_OPT_WALK == 2
        RT_WALK_CLOCK(cen, 6, t_trip);
        const unsigned long long t_s = __builtin_amdgcn_s_memtime();
        {
            // What MORE RAYS THAN LANES would make of this very walk phase, measured on the frame: this lane's steps
            // of the trip (a leaf step weighs four pair steps: 212 against 54 instructions), the phase's length as executed (the
            // slowest lane), and its length if lanes l and l + 32 (l, l + 16, l + 32, l + 48) were ONE lane walking their rays one
            // after the other -- two (four) rays per lane with a free, instant switch.  counters[0..3] = sums over all trips of:
            // the slowest lane, all lanes' steps, the slowest pair, the slowest four.
            uint32_t mine = (uint32_t)(cen[1] - emu_p0) + 4u * (uint32_t)(cen[3] - emu_l0);
            uint32_t two = mine + (uint32_t)__shfl_xor((int)mine, 32, 64);
            uint32_t four = two + (uint32_t)__shfl_xor((int)two, 16, 64);
            uint32_t m1 = mine, m2 = two, m4 = four, sum = mine;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                m1 = max(m1, (uint32_t)__shfl_xor((int)m1, off, 64));
                m2 = max(m2, (uint32_t)__shfl_xor((int)m2, off, 64));
                m4 = max(m4, (uint32_t)__shfl_xor((int)m4, off, 64));
                sum += (uint32_t)__shfl_xor((int)sum, off, 64);
            }
            if (lane == 0) {
                emu[0] += m1;
                emu[1] += sum;
                emu[2] += m2;
                emu[3] += m4;
            }
        }
#endif
#endif      // !RT_OPT_RAYS2
#if RT_OPT_RAYS2
#if RT_OPT_WALK == 2
        RT_WALK_CLOCK(cen, 6, t_trip);
        const unsigned long long t_s = __builtin_amdgcn_s_memtime();
#endif
        // ---- S, once for the ray at hand and once for the other (every walk has ended: whoever has a ray shades) ----
#pragma nounroll
        for (int side = 0; side < 2; ++side) {
        const bool finished = ctl.st() == kNew && s >= s_end_now;
#else
        {
#endif
        // ---- S: lanes whose walk has ended, once enough of them wait ----
        const bool ready = (W.cur == kWalkDone) & !finished;
        const unsigned long long br = __builtin_amdgcn_ballot_w64(ready);
        const unsigned long long bw = __builtin_amdgcn_ballot_w64(W.cur != kWalkDone);
        const bool go = (__popcll(br) >= P.regen_gate) || (bw == 0ull);
        if (ready && go) {
            RT_WALK_COUNT(4, __builtin_amdgcn_ballot_w64(true));
            bool path_done = false;
            int start = 0;                  // the ray this lane starts at the end of the phase: 0 none, 1 closest hit, 2 shadow
            if (ctl.st() == kShadow) {
                // ---- the shadow ray of light ctl.light() - 1 has its answer, .cl:297-301 ----
                const bool blocked = W.idx < n;
                c_tests += blocked ? W.idx + 1u : n;
                if ((int)c_tests < 0) {                                             // (rare: the 64-bit sum lives in LDS, the lane keeps 31 bits of it)
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
                if (!(W.far < 1e20f)) {
                    path_done = true;                                              // miss, .cl:327-330
                } else {
                    const float4 ge = s_slots[W.slot];
                    float4 em4, co4;
                    // the hit sphere's material by the SLOT the walk ended on (the blob's material sections are in slot order):
                    // one round trip to L2, where the scene index first and the record by index after it were two
                    em4 = P.bvh.blob[P.bvh.emis_at + W.slot];
                    co4 = P.bvh.blob[P.bvh.emis_at + n_slots + W.slot];
                    const V3 em = mk(em4.x, em4.y, em4.z);
                    const V3 col = mk(co4.x, co4.y, co4.z);
                    const int refl = __float_as_int(em4.w);
                    const V3 hp = add(o, scale(d, W.far));                         // .cl:338-340
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
                            float cos2t = cos2t_of(nnt, ddn);
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
                                float Pr = roulette_p(Re);
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
            // ---- next-event estimation, .cl:249-303 (the lights one by one, each with its two draws), then the cosine-weighted
            //      bounce, .cl:383-411 (two draws as well).  Either begins with two random numbers, the sine and cosine of 2 pi
            //      times one of them and the square root of a value formed from the other: that part is ONE section for the lanes
            //      about to sample a light and the lanes about to bounce.  Per pixel the operations and their order are the
            //      reference's (sample_light of rt_trace.inc.h and its bounce, term for term) ----
            while (ctl.st() == kLights) {
                const bool bounce = ctl.light() == n_lights;
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
                    float wo = dot_decision(sd, us);
                    const float wi = dot_decision(sd, nl);
                    if (!(wo > 0.f) && wi > 0.f) {                                 // .cl:283-296: this side of the light, facing it
                        wo = -wo;
                        // ---- shadow ray, any hit, .cl:234-247: the large spheres at the end of this phase, the tree in the trips to come ----
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
            if (ctl.st() == kNew && start == 0 && s < s_end_now) {
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
                    uint32_t xy_now = RT_XY_;
                    asm volatile("; pixel coordinates unpacked per sample" : "+v"(xy_now));       // (as in rt_trace.inc.h: not hoisted into two more registers)
                    float kcx = ((float)(xy_now & 0xffffu) + j1) * inv_w - 0.5f;
                    float kcy = ((float)(xy_now >> 16) + j2) * inv_h - 0.5f;
                    V3 rd = mk(cam_x.x * kcx + cam_y.x * kcy + cam_d.x, cam_x.y * kcx + cam_y.y * kcy + cam_d.y,
                               cam_x.z * kcx + cam_y.z * kcy + cam_d.z);
                    o = add(scale(rd, 0.1f), cam_o);
                    d = unit(rd);
                    thr = mk(1.f, 1.f, 1.f);
                    rad = mk(0.f, 0.f, 0.f);
                    ctl.new_path();
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
#if RT_OPT_WALK == 2
                cen[9] += shadow ? (first < n_always ? first + 1u : n_always) : n_always;
#endif
                if (shadow) {
                    W.idx = first < n_always ? s_index[first] : n;
                } else {
                    W.far = t;
                    W.slot = slot;
                    ctl.set_st(kClosest);
                }
                R = bvh_ray(s_hdr, o, d);
                W.cur = bvh_root(s_hdr);           // (from the header the ray set-up has just read: whoever built the tree put it there)
                W.sp = 0;
            }
        }
#if RT_OPT_RAYS2
        swap_rays();
#endif
        }       // (the shade phase: once, or once per ray of the lane)
#if RT_OPT_WALK == 2
        RT_WALK_CLOCK(cen, 7, t_s);
#endif
    }
#if RT_OPT_WALK == 2
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
    if (lane == 0)
        for (int k = 0; k < 4; ++k) atomicAdd(&P.counters[k], emu[k]);
    for (int k = 10; k < 13; ++k) {
        unsigned long long h = hist[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off, 64);
        if (lane == 0) atomicAdd(&P.counters[k == 12 ? 31 : 8 + k], h);
    }
#endif

    // ---- epilogue: as in rt_trace.inc.h ----
    const __attribute__((address_space(4))) LaunchParams *qp =
        (const __attribute__((address_space(4))) LaunchParams *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; epilogue arguments re-read" : "+s"(qp));
    const __attribute__((address_space(4))) LaunchParams &Q = *qp;
    uint32_t n_done = 0u;
#if RT_OPT_RAYS2
#pragma nounroll
    for (int side = 0; side < 2; ++side) {
#else
    {
#endif
    const bool valid_e = s_end_now != Q.first_sample;       // (s_end was first_sample + n_samples for the lanes that own a pixel)
    n_done += valid_e ? (uint32_t)Q.n_samples : 0u;
    if (valid_e && Q.n_samples > 0) {
        uint32_t xy_e = RT_XY_;
        asm volatile("; pixel coordinates unpacked after the loop" : "+v"(xy_e));
        const int xe = (int)(xy_e & 0xffffu), ye = (int)(xy_e >> 16);
        int le = tile_by * kTileH + ((int)(threadIdx.x & 63u) >> 3);
        const size_t gid = (size_t)ye * (size_t)Q.w + (size_t)xe;
        const size_t ci = (size_t)(Q.h - ye - 1) * (size_t)Q.w + (size_t)xe;
        float *colors = Q.colors;
        colors[3 * ci] = acc.x;
        colors[3 * ci + 1] = acc.y;
        colors[3 * ci + 2] = acc.z;
        if (!(Q.skip_pixels & 1))
            Q.pixels[(size_t)le * (size_t)Q.w + (size_t)xe] =
                (uint32_t)(to_int(acc.x) | (to_int(acc.y) << 8) | (to_int(acc.z) << 16));
        *reinterpret_cast<uint2 *>(Q.seeds + 2 * gid) = make_uint2(s0, s1);
    }
#if RT_OPT_RAYS2
    swap_rays();
#endif
    }       // (the lane's pixel, or its two)
#undef RT_XY_
#undef RT_SWAP_
#undef RT_SWAP_V3_
    uint32_t t_samples = wave_sum(n_done);
    uint32_t t_closest = wave_sum(c_closest);
    uint32_t t_shadow = wave_sum(c_shadow);
    uint32_t t_draws = wave_sum(c_draws);
    unsigned long long tests64 = (unsigned long long)c_tests + (unsigned long long)c_closest * n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tests64 += __shfl_xor(tests64, off, 64);
    if (lane == 0) atomicMax(&s_tile_cost, (unsigned)(__builtin_amdgcn_s_memrealtime() - s_wg_t0));      // (as in rt_trace.inc.h)
    if (lane == 0) {
        atomicAdd(&s_stat[0], (unsigned long long)t_samples);
        atomicAdd(&s_stat[1], (unsigned long long)t_closest);
        atomicAdd(&s_stat[2], (unsigned long long)t_shadow);
        atomicAdd(&s_stat[3], tests64);
        atomicAdd(&s_stat[4], (unsigned long long)t_draws);
    }
    __syncthreads();
    if (tid == 5 && Q.tile_cost) Q.tile_cost[tile_id] = (Q.skip_pixels & 2) ? Q.tile_cost[tile_id] + s_tile_cost : s_tile_cost;       // (as in rt_trace.inc.h)
    if (tid < 5) atomicAdd(&Q.stats[(block_linear % (unsigned)kStatReplicas) * 8u + (unsigned)tid], s_stat[tid]);
}

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
        uint32_t cur = bvh_root(s_hdr), w_slot = 0, w_idx;
        int sp = 0;
        float w_far;
        // the two kinds of ray diverge here; each sweep's ballots see the lanes of its own kind
        if (shadow) {
            const uint32_t first_large = sweep_any(s_slots, n_always, o, d, ra.w, roots);
            w_idx = first_large < n_always ? g_index[first_large] : n;
            w_far = ra.w;
            walk_pairs(s_pairs, s_slots, g_index, s_stack + tid, kThreads, n_always, o, d, R, true, 3, cur, sp, w_far, w_idx, w_slot, nullptr, nullptr);
            const uint32_t ref = sweep_any(P.scene.geom, n, o, d, ra.w, roots);
            res = make_uint4(w_idx, 0u, ref, 0u);
        } else {
            float t = 1e20f;
            uint32_t slot = 0;
            sweep_closest(s_slots, n_always, o, d, t, slot, roots);
            w_far = t;
            w_slot = slot;
            w_idx = 0xffffffffu;
            walk_pairs(s_pairs, s_slots, g_index, s_stack + tid, kThreads, n_always, o, d, R, false, 3, cur, sp, w_far, w_idx, w_slot, nullptr, nullptr);
            w_idx = g_index[w_slot];                    // (read for a miss too: slot 0 then; the result below ignores it)
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
