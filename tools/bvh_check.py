#!/usr/bin/env python3
"""The hierarchy of large scenes (csrc/rt_device.h BvhTables), checked on the GPU:

  structure   the tables the device (or, beyond 8192 tree spheres, the host) built, read back and walked on the host:
              from the root pair every leaf is reached exactly once, the leaves hold every tree sphere exactly once,
              every sphere lies inside the boxes of its leaf and of all its ancestors, every child knows the lowest
              scene index below it, the always list keeps scene order
  rays        rays through the walk AND the plain sweep, one lane per ray (rt_debug_walk_rays): camera rays and bounce
              rays of the scene itself here; tests/test_gpu_bvh.py adds the adversarial ones
  parity      the shipped instance against the oracle, and timing with the hierarchy on / off
  census      what the walk executes (instance rt_trace_parity_pairs_census): pair steps, leaf steps, lanes taking part

    python tools/bvh_check.py [--quick] [--census] [--timing-only]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytracing_simple_amd import api, host, scenes  # noqa: E402


def read_bvh(ctx):
    counts = (C.c_uint32 * 4)()
    ctx._check(ctx._lib.rt_debug_read_bvh(ctx._h, None, 0, counts))
    n_always, n_leaves, depth, root = list(counts)
    n_slots = n_always + 8 * n_leaves
    if n_leaves == 0:
        return None
    at_index = 2 + n_slots
    at_pairs = at_index + (n_slots + 3) // 4
    at_emis = at_pairs + 4 * (n_leaves - 1)
    n4 = at_emis + 2 * n_slots                          # ... | pairs | material records by slot (emission + reflection, colour + radius)
    blob = np.zeros(4 * n4, np.float32)
    ctx._check(ctx._lib.rt_debug_read_bvh(ctx._h, blob.ctypes.data_as(C.c_void_p), n4, counts))
    b4 = blob.reshape(n4, 4)
    return {"hdr": b4[:2], "slots": b4[2:at_index], "index": blob[4 * at_index:].view(np.uint32)[:n_slots], "pairs": b4[at_pairs:at_emis],
            "emis": b4[at_emis:at_emis + n_slots], "colr": b4[at_emis + n_slots:at_emis + 2 * n_slots],
            "n_always": n_always, "n_leaves": n_leaves, "stack_depth": depth, "n_slots": n_slots, "root": root}


def read_packed(ctx):
    """The packed pair table (rt_debug_read_packed_pairs): {"r0", "scale" (float32[3]), "words" (uint32[n_pairs, 8])} or None."""
    n = C.c_uint32()
    ctx._check(ctx._lib.rt_debug_read_packed_pairs(ctx._h, None, 0, C.byref(n)))
    if n.value == 0:
        return None
    raw = np.zeros(8 + 8 * n.value, np.uint32)
    ctx._check(ctx._lib.rt_debug_read_packed_pairs(ctx._h, raw.ctypes.data_as(C.c_void_p), raw.nbytes, C.byref(n)))
    return {"r0": raw[0:3].view(np.float32), "scale": raw[4:7].view(np.float32), "words": raw[8:].reshape(n.value, 8)}


def check_packed(b, pk, low_shift=2):
    """The packed table against the pairs it was made from (rt_bvh.hip rt_bvh_pack_pairs_kernel): every packed box CONTAINS its pair's box
    (in exact arithmetic: plane = r0 + q * scale), by no more than three cells a side; references equal; the packed lowest scene index is a
    lower bound of the pair's, at most 2^shift - 1 below it (or saturated).  Returns a list of complaints."""
    bad = []
    pr = b["pairs"].reshape(-1, 4, 4)
    w = pk["words"]
    if len(w) != len(pr):
        return ["%d packed records for %d pairs" % (len(w), len(pr))]
    r0, sc = pk["r0"].astype(np.float64), pk["scale"].astype(np.float64)
    lo16, hi16 = (lambda v: (v & 0xffff).astype(np.float64)), (lambda v: (v >> 16).astype(np.float64))
    # words: l0x|l0y, l0z|h0x, h0y|h0z, l1x|l1y, l1z|h1x, h1y|h1z, ref0|ref1, low0|low1
    q_lo = [np.stack([lo16(w[:, 0]), hi16(w[:, 0]), lo16(w[:, 1])], 1), np.stack([lo16(w[:, 3]), hi16(w[:, 3]), lo16(w[:, 4])], 1)]
    q_hi = [np.stack([hi16(w[:, 1]), lo16(w[:, 2]), hi16(w[:, 2])], 1), np.stack([hi16(w[:, 4]), lo16(w[:, 5]), hi16(w[:, 5])], 1)]
    for side in (0, 1):
        lo, hi = pr[:, 2 * side, :3].astype(np.float64), pr[:, 2 * side + 1, :3].astype(np.float64)
        plo, phi = r0 + q_lo[side] * sc, r0 + q_hi[side] * sc
        if np.any(plo > lo) or np.any(phi < hi):
            bad.append("side %d: %d packed boxes do not contain their pair's" % (side, int(np.sum(np.any(plo > lo, 1) | np.any(phi < hi, 1)))))
        slack = np.maximum(lo - plo, phi - hi) / np.maximum(sc, 1e-300)
        inner = (q_lo[side] > 0) & (q_hi[side] < 65535)
        if np.any(slack[inner] > 3.01):
            bad.append("side %d: a packed plane lies %.2f cells off its pair's" % (side, float(slack[inner].max())))
        ref = pr[:, 2 * side, 3].view(np.uint32) & 0xffff
        got = (w[:, 6] & 0xffff) if side == 0 else (w[:, 6] >> 16)
        if np.any(ref != got):
            bad.append("side %d: references differ" % side)
        low = pr[:, 2 * side + 1, 3].view(np.uint32).astype(np.int64)
        lq = ((w[:, 7] & 0xffff) if side == 0 else (w[:, 7] >> 16)).astype(np.int64)
        bound = lq << low_shift
        if np.any(bound > low) or np.any((low - bound >= (1 << low_shift)) & (lq != 0xffff)):
            bad.append("side %d: the packed lowest scene index is not a tight lower bound" % side)
    return bad


def sum_of_box_areas(b):
    """Surface areas of every child box of every pair, summed: what a random ray's expected number of box visits goes with."""
    pr = b["pairs"].reshape(-1, 4, 4)
    total = 0.0
    for side in (0, 2):
        d = pr[:, side + 1, :3].astype(np.float64) - pr[:, side, :3].astype(np.float64)
        total += float((d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]).sum())
    return total


def first_of_equals(sph):
    """Scene indices the hierarchy holds: every record but those that repeat an EARLIER finite record bit for bit in centre and radius^2
    (rt_bvh.hip mark_duplicates: the reference's loops keep the first of equals, so a repeated record is never a ray's answer)."""
    rad = sph["rad"].astype(np.float32)
    key = np.concatenate([np.ascontiguousarray(sph["p"]).astype(np.float32).view(np.uint32).reshape(-1, 3), (rad * rad).view(np.uint32).reshape(-1, 1)], axis=1)
    finite = np.isfinite(rad) & np.all(np.isfinite(np.ascontiguousarray(sph["p"]).astype(np.float32)), axis=1) & (np.abs(rad) <= 3.0e38)
    seen, keep = set(), []
    for i in range(len(sph)):
        k = tuple(int(v) for v in key[i])
        if finite[i] and k in seen:
            continue
        if finite[i]:
            seen.add(k)
        keep.append(i)
    return keep


def check_structure(sph, b):
    """Host-side walk of the tables the library built; returns a list of complaints (empty = fine)."""
    bad = []
    n = len(sph)
    na, nl = b["n_always"], b["n_leaves"]
    leaf_size = (b["n_slots"] - na) // max(nl, 1)
    idx = b["index"]
    # always list: scene order, records equal
    al = idx[:na]
    if not np.all(np.diff(al.astype(np.int64)) > 0):
        bad.append("always list not in scene order")
    tree_idx = idx[na:]
    real = tree_idx[tree_idx != 0xffffffff]
    if sorted(list(al) + list(real)) != first_of_equals(sph):
        bad.append("slots do not hold every sphere (the first of bit-equal records) exactly once")
    rad = sph["rad"].astype(np.float32)
    p = np.ascontiguousarray(sph["p"]).astype(np.float32)
    for j in range(b["n_slots"]):
        ix = idx[j]
        if ix == 0xffffffff:
            if not np.all(np.isnan(b["slots"][j])):
                bad.append(f"padding slot {j} is not NaN")
            continue
        want = np.array([p[ix, 0], p[ix, 1], p[ix, 2], np.float32(rad[ix]) * np.float32(rad[ix])], np.float32)
        if not np.array_equal(want.view(np.uint32), b["slots"][j].view(np.uint32)):
            bad.append(f"slot {j} != record {ix}")
        # the material records of the slot: { emission, bits(refl) }, { colour, radius } of that very sphere
        e, c = np.ascontiguousarray(sph["e"][ix]).astype(np.float32), np.ascontiguousarray(sph["c"][ix]).astype(np.float32)
        want_e = np.concatenate([e.view(np.uint32), np.array([sph["refl"][ix]], np.int32).view(np.uint32)])
        want_c = np.concatenate([c.view(np.uint32), np.array([rad[ix]], np.float32).view(np.uint32)])
        if not (np.array_equal(want_e, b["emis"][j].view(np.uint32)) and np.array_equal(want_c, b["colr"][j].view(np.uint32))):
            bad.append(f"material records of slot {j} != sphere {ix}")
    # the sibling pairs: from the root every leaf is reached exactly once, every child box holds the
    # spheres below it and knows their lowest scene index
    LEAF = 0x8000
    pr = b["pairs"]
    seen_leaves, seen_pairs = [], set()

    deepest = [1]

    def below(ref, level=1):
        """(spheres below ref), walking the pairs"""
        deepest[0] = max(deepest[0], level)
        if ref & LEAF:
            lf = ref & (LEAF - 1)
            seen_leaves.append(lf)
            members = [int(i) for i in tree_idx[leaf_size * lf:leaf_size * lf + leaf_size]]
            real_n = sum(1 for i in members if i != 0xffffffff)
            if real_n == 0 or any(i != 0xffffffff for i in members[real_n:]):
                bad.append(f"leaf {lf}: empty, or padding before a record")
            return [i for i in members if i != 0xffffffff]
        if ref in seen_pairs or ref >= nl - 1:
            bad.append(f"pair {ref} reached twice or out of range")
            return []
        seen_pairs.add(ref)
        out = []
        for side in (0, 1):
            A, B = pr[4 * ref + 2 * side], pr[4 * ref + 2 * side + 1]
            child = int(A[3:4].view(np.uint32)[0])
            members = below(child, level + 1)
            if members:
                m = np.array(members)
                ar = np.abs(rad[m]).astype(np.float64)
                if np.any(A[:3] > (p[m].astype(np.float64) - ar[:, None]).min(0)) or np.any(B[:3] < (p[m].astype(np.float64) + ar[:, None]).max(0)):
                    bad.append(f"pair {ref} side {side}: box does not hold its spheres")
                if int(B[3:4].view(np.uint32)[0]) != m.min():
                    bad.append(f"pair {ref} side {side}: lowest index wrong")
            else:
                bad.append(f"pair {ref} side {side}: nothing below")
            out += members
        return out

    import sys as _sys
    _sys.setrecursionlimit(10000)
    # the root's pair comes with the tables: the device build halves every range of leaves (root = pair nl // 2 - 1), the
    # host build of a full upload cuts by surface area
    root = b["root"]
    if nl == 1 and root != LEAF:
        bad.append("a tree of one leaf has no root pair")
    everything = below(root)
    if b["stack_depth"] < deepest[0] or b["stack_depth"] > deepest[0] + 4:      # (a tree shaped on the device sizes the stacks for its depth BOUND: up to three levels more)
        bad.append(f"stack depth {b['stack_depth']} for a tree of {deepest[0]} levels")
    if sorted(seen_leaves) != list(range(nl)):
        bad.append("the pairs do not reach every leaf exactly once")
    if sorted(everything) != sorted(int(i) for i in real):
        bad.append("the pairs do not reach every tree sphere exactly once")
    return bad


def counters_raw(ctx):
    out = (C.c_ulonglong * 32)()
    ctx._check(ctx._lib.rt_debug_counters_raw(ctx._h, out))
    return list(out)


def scene_rays(sph, cam, w, h, n, seed=1):
    """Rays a render of this scene traces: camera rays through random pixels (closest hit) and, from random points on
    the spheres' surfaces, cosine-ish bounce rays (closest hit) and rays towards the lights (shadow rays, bounded)."""
    rng = np.random.default_rng(seed)
    cam = np.asarray(cam, np.float32)
    orig, cdir, cx, cy = cam[0:3], cam[6:9], cam[9:12], cam[12:15]
    p = np.ascontiguousarray(sph["p"]).astype(np.float64)
    r = np.abs(sph["rad"].astype(np.float64))
    ok = np.isfinite(p).all(1) & np.isfinite(r) & (r > 0)
    lights = np.nonzero(ok & ((sph["e"][:, 0] != 0) | (sph["e"][:, 2] != 0)))[0]
    small = np.nonzero(ok & (r < 500))[0]
    rays = np.zeros((n, 8), np.float32)
    for i in range(n):
        kind = i % 3
        if kind == 0 or len(small) == 0:
            kx, ky = rng.uniform(-0.5, 0.5, 2)
            d = cx * kx + cy * ky + cdir
            o = orig + 0.1 * d
            d = d / np.linalg.norm(d)
            tmax, shadow = 1e20, 0
        else:
            j = int(small[rng.integers(0, len(small))])
            nrm = rng.normal(0, 1, 3); nrm /= np.linalg.norm(nrm)
            o = p[j] + nrm * r[j]
            if kind == 1 or len(lights) == 0:
                d = nrm + rng.normal(0, 0.6, 3); d /= np.linalg.norm(d)
                tmax, shadow = 1e20, 0
            else:
                k = int(lights[rng.integers(0, len(lights))])
                u = rng.normal(0, 1, 3); u /= np.linalg.norm(u)
                to = p[k] + u * r[k] - o
                tmax = float(np.linalg.norm(to)) - 0.01
                d = to / max(np.linalg.norm(to), 1e-30)
                shadow = 1
        rays[i, 0:3], rays[i, 3], rays[i, 4:7] = o, tmax, d
        rays[i, 7:8].view(np.uint32)[0] = shadow
    return rays


def walk_rays(ctx, rays):
    """rt_debug_walk_rays: out[i] = the walk's answer (2 words), then the sweep's (2 words)"""
    out = np.zeros((len(rays), 4), np.uint32)
    with np.errstate(all="ignore"):
        ctx._check(ctx._lib.rt_debug_walk_rays(ctx._h, rays.ctypes.data_as(C.c_void_p), len(rays), out.ctypes.data_as(C.c_void_p)))
    return out


def agreement(sph, cam, w, h, n_rays=120000, bvh_min=1):
    """Rays of the scene itself through the walk and through the plain sweep: how many answers differ (must be 0)."""
    sph = api.as_spheres(sph)
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, bvh_min, 152 * 1024))
        ctx.set_scene(sph)
        b = read_bvh(ctx)
        if b is None:
            return None
        rays = scene_rays(sph, cam, w, h, n_rays)
        out = walk_rays(ctx, rays)
    shadow = rays[:, 7].view(np.uint32) != 0
    differ = (out[:, 0] != out[:, 2]) | (out[:, 1] != out[:, 3])
    return {"closest_rays": int((~shadow).sum()), "closest_differ": int((differ & ~shadow).sum()), "closest_hits": int(((out[:, 2] != 0xffffffff) & ~shadow).sum()),
            "shadow_rays": int(shadow.sum()), "shadow_differ": int((differ & shadow).sum()), "shadow_blocked": int(((out[:, 2] < len(sph)) & shadow).sum()),
            "tree": [b["n_always"], b["n_leaves"], b["stack_depth"]]}


def census_walk(sph, cam, w, h, spp, steps=0, gate=0):
    """instance rt_trace_parity_pairs_census: what the walk executes -- pair steps (two box tests each) and leaf steps
    (kBvhLeaf sphere tests each) per wavefront and per lane, shade phases, loop trips, clock shares"""
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 152 * 1024))
        ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, steps, gate, 1))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        ctx.set_mode(api.instance_mode("rt_trace_parity_pairs_census"))
        ctx.render_pass(spp)
        c = counters_raw(ctx)[20:29]
        st = ctx.stats()
        b = read_bvh(ctx)
    rays = st["closest_rays"] + st["shadow_rays"]
    return {"rays": rays, "pair_steps_lanes": c[1], "pair_steps_waves": c[0], "leaf_steps_lanes": c[3], "leaf_steps_waves": c[2],
            "always_spheres": b["n_always"], "node_tests_per_ray": round(c[1] / rays, 1), "lanes_per_node_step": round(c[1] / max(c[0], 1), 1),
            "leaf_visits_per_ray": round(c[3] / rays, 2), "lanes_per_leaf_step": round(c[3] / max(c[2], 1), 1),
            "shade_phases": c[4], "lanes_per_shade_phase": round(c[5] / max(c[4], 1), 1), "loop_trips": c[8],
            "node_steps_per_trip": round(c[0] / max(c[8], 1), 1), "clock_share_walk": round(c[6] / max(c[6] + c[7], 1), 3)}


def timed(sph, cam, w, h, spp, bvh_min, reps=3, mode=api.RT_MODE_PARITY, walk=(0, 0, 0), ratio=0, lds_limit=0):
    with api.RtContext(w, h, diag=True) as ctx:
        ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, bvh_min, lds_limit))
        ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, *walk))
        if ratio:
            ctx._check(ctx._lib.rt_debug_set_walk_round(ctx._h, ratio))
        ctx.set_scene(sph)
        ctx.set_camera(cam)
        ctx.set_mode(mode)
        best = None
        px = None
        for _ in range(reps):
            ctx.reset()
            px = ctx.render_pass(spp)
            ms = ctx.stats()["last_kernel_ms"]
            best = ms if best is None else min(best, ms)
        st = ctx.stats()
        st["pick"] = ctx._lib.rt_debug_bvh_pick(ctx._h)
        return best, px, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--boxes", action="store_true", help="timing on enclosed all-specular scenes instead of the open ones")
    ap.add_argument("--timing-only", action="store_true")
    ap.add_argument("--census", action="store_true", help="only the step census of the walk (c3 and c5 at quarter size)")
    args = ap.parse_args()
    report = {}
    if args.census:
        for name, mk, (w, h, spp) in [("c3", lambda: scenes.random_spheres(1024), (480, 270, 16)),
                                      ("mirror_box_256", lambda: scenes.mirror_box(256), (480, 270, 16))]:
            sph, orig, target = mk()
            for steps, gate in ((64, 16), (16, 16)):
                print("census_pairs", name, steps, gate, json.dumps(census_walk(sph, host.compute_camera(orig, target, w, h), w, h, spp, steps, gate)), flush=True)
        return 0
    if args.timing_only:
        return timing(args, report)
    # --- structure ---
    makers = {"random_1024": lambda: scenes.random_spheres(1024), "random_96": lambda: scenes.random_spheres(96),
              "mirror_box_64": lambda: scenes.mirror_box(64), "demo_plus_16": lambda: scenes.demo_plus(16),
              "random_37": lambda: scenes.random_spheres(37)}
    for name, mk in makers.items():
        sph, orig, target = mk()
        sph = api.as_spheres(sph)
        with api.RtContext(64, 64, diag=True) as ctx:
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 0))
            ctx.set_scene(sph)
            b = read_bvh(ctx)
            bad = check_structure(sph, b) if b else ["no hierarchy"]
            report["structure_" + name] = {"always": b["n_always"], "leaves": b["n_leaves"], "stack_depth": b["stack_depth"], "complaints": bad[:5],
                                           "hdr": [float(v) for v in b["hdr"].ravel()[:7]]}
            print("structure", name, report["structure_" + name], flush=True)
            if bad:
                print(json.dumps(report))
                return 1
    # --- agreement ---
    for name, mk, (w, h, spp) in [("random_1024", lambda: scenes.random_spheres(1024), (192, 108, 4)),
                                  ("mirror_box_64", lambda: scenes.mirror_box(64), (128, 96, 8)),
                                  ("demo_plus_16", lambda: scenes.demo_plus(16), (128, 96, 8)),
                                  ("random_96", lambda: scenes.random_spheres(96), (128, 96, 8))]:
        sph, orig, target = mk()
        cam = host.compute_camera(orig, target, w, h)
        r = agreement(sph, cam, w, h)
        report["agreement_" + name] = r
        print("agreement", name, r, flush=True)
        if r is None or r["closest_differ"] or r["shadow_differ"]:
            print(json.dumps(report))
            return 1
    # --- parity against the oracle (small) and timing on / off ---
    import _oracle as O
    for name, mk, (w, h, spp) in [("random_1024", lambda: scenes.random_spheres(1024), (96, 64, 2)),
                                  ("mirror_box_64", lambda: scenes.mirror_box(64), (64, 64, 4))]:
        sph, orig, target = mk()
        cam = host.compute_camera(orig, target, w, h)
        want = O.render(sph, cam, w, h, spp)
        for form in (1,):
          with api.RtContext(w, h, diag=True) as ctx:
            ctx._check(ctx._lib.rt_debug_set_bvh(ctx._h, 1, 0))
            ctx._check(ctx._lib.rt_debug_set_walk(ctx._h, 0, 0, form))
            ctx.set_scene(sph)
            ctx.set_camera(cam)
            px = ctx.render_pass(spp)
            st = ctx.stats()
            same = bool(np.array_equal(px, want["pixels"]) and np.array_equal(ctx.read_colors().view(np.uint32), want["colors"].view(np.uint32))
                        and np.array_equal(ctx.read_seeds(), want["seeds"]))
            cnt = (st["samples"], st["closest_rays"], st["shadow_rays"], st["sphere_tests"], st["rng_draws"]) == \
                  (want["stats"]["samples"], want["stats"]["closest_calls"], want["stats"]["shadow_calls"], want["stats"]["sphere_tests"], want["stats"]["rng_draws"])
          report[f"oracle_{name}_form{form}"] = {"frame_equal": same, "counters_equal": cnt}
          print("oracle", name, "form", form, report[f"oracle_{name}_form{form}"], flush=True)
          if not (same and cnt):
              return 1
    return timing(args, report)


def timing(args, report):
    full = [("c3_random_1024", lambda: scenes.random_spheres(1024), (1920, 1080, 16)),
            ("random_512", lambda: scenes.random_spheres(512), (1920, 1080, 16)),
            ("random_256", lambda: scenes.random_spheres(256), (1920, 1080, 16)),
            ("random_128", lambda: scenes.random_spheres(128), (1920, 1080, 16)),
            ("mirror_box_256", lambda: scenes.mirror_box(256), (1920, 1080, 16)),
            ("c5_mirror_box_64", lambda: scenes.mirror_box(64), (1920, 1080, 16))]
    if args.boxes:
        full = [("mirror_box_512", lambda: scenes.mirror_box(512), (960, 540, 16)), ("mirror_box_1024", lambda: scenes.mirror_box(1024), (960, 540, 16)),
                ("mirror_box_2048", lambda: scenes.mirror_box(2048), (960, 540, 16)), ("random_2048", lambda: scenes.random_spheres(2048), (960, 540, 16))]
    if args.quick:
        full = [(n, m, (w // 4, h // 4, s)) for n, m, (w, h, s) in full]
    for name, mk, (w, h, spp) in full:
        sph, orig, target = mk()
        cam = host.compute_camera(orig, target, w, h)
        t_off, px_off, st_off = timed(sph, cam, w, h, spp, 0)
        t_on, px_on, st_on = timed(sph, cam, w, h, spp, 1, walk=(0, 0, 1))
        t_auto, px_auto, st_auto = timed(sph, cam, w, h, spp, 1, walk=(0, 0, 0))
        sweep = {}
        for steps, gate in ((32, 16), (64, 16), (128, 16), (64, 32)):
            tt, pp, _ = timed(sph, cam, w, h, spp, 1, reps=2, walk=(steps, gate, 1))
            sweep[f"{steps}/{gate}"] = round(tt, 2) if np.array_equal(pp, px_off) else "FRAME DIFFERS"
        rays = st_on["samples"] + st_on["shadow_rays"]
        report["timing_" + name] = {"plain_ms": round(t_off, 3), "hierarchy_ms": round(t_on, 3), "measured_choice_ms": round(t_auto, 3), "measured_choice": st_auto.get("pick"),
                                    "frames_equal": bool(np.array_equal(px_off, px_on) and np.array_equal(px_off, px_auto)), "steps/gate_ms": sweep,
                                    "counters_equal": st_off["sphere_tests"] == st_on["sphere_tests"],
                                    "Mray_s_hierarchy": round(rays / t_on / 1e3, 1), "Mray_s_plain": round(rays / t_off / 1e3, 1)}
        print("timing", name, report["timing_" + name], flush=True)
    print(json.dumps(report))
    return 0


if __name__ == "__main__":
    sys.exit(main())
