#!/usr/bin/env python3
"""Surface-area cost of the hierarchy under three ways of deciding who sits in which leaf (the tree's shape -- leaves of 8,
ranges split in the middle -- is the library's in the first two): one sort along a Morton curve, a top-down median split
along the longest axis of the centres' box (what csrc/rt_bvh.hip does), and a binned surface-area heuristic.  CPU only.
    python tools/tree_quality.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_simple_amd import scenes, api

def area(lo, hi):
    d = np.maximum(hi - lo, 0)
    return 2 * (d[0]*d[1] + d[1]*d[2] + d[0]*d[2])

def morton_order(c):
    lo, hi = c.min(0), c.max(0)
    q = np.clip(((c - lo) / np.maximum(hi - lo, 1e-30) * 1023), 0, 1023).astype(np.uint32)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    code = (spread(q[:, 0]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 2])
    return np.argsort(code, kind="stable"), code

def build_count(order, c, r, leaf=8):
    n = len(order)
    nl = (n + leaf - 1) // leaf
    leaves = [order[i*leaf:(i+1)*leaf] for i in range(nl)]
    inner, leafc = 0.0, 0.0
    def rec(a, b):
        nonlocal inner, leafc
        idx = np.concatenate(leaves[a:b])
        lo, hi = (c[idx] - r[idx, None]).min(0), (c[idx] + r[idx, None]).max(0)
        A = area(lo, hi)
        if b - a == 1:
            leafc += A * len(idx)
            return A
        inner += A
        m = (a + b) // 2
        rec(a, m); rec(m, b)
        return A
    root = rec(0, nl)
    return inner / root, leafc / root

def build_median(c, r, leaf=8):
    inner, leafc = 0.0, 0.0
    def rec(idx):
        nonlocal inner, leafc
        lo, hi = (c[idx] - r[idx, None]).min(0), (c[idx] + r[idx, None]).max(0)
        A = area(lo, hi)
        if len(idx) <= leaf:
            leafc += A * len(idx)
            return A
        inner += A
        ext = c[idx].max(0) - c[idx].min(0)
        ax = int(np.argmax(ext))
        o = idx[np.argsort(c[idx, ax], kind="stable")]
        # split so that the left part is a multiple of `leaf` nearest the middle (keeps leaves full)
        half = len(o) // 2
        half = max(leaf, (half + leaf // 2) // leaf * leaf) if len(o) > 2 * leaf else half
        rec(o[:half]); rec(o[half:])
        return A
    root = rec(np.arange(len(c)))
    return inner / root, leafc / root

def build_sah(c, r, leaf=8, bins=16):
    inner, leafc = 0.0, 0.0
    def rec(idx):
        nonlocal inner, leafc
        lo, hi = (c[idx] - r[idx, None]).min(0), (c[idx] + r[idx, None]).max(0)
        A = area(lo, hi)
        if len(idx) <= leaf:
            leafc += A * len(idx)
            return A
        inner += A
        best = None
        for ax in range(3):
            o = idx[np.argsort(c[idx, ax], kind="stable")]
            for k in range(1, bins):
                s = len(o) * k // bins
                if s == 0 or s == len(o): continue
                L, Rr = o[:s], o[s:]
                aL = area((c[L]-r[L,None]).min(0), (c[L]+r[L,None]).max(0)); aR = area((c[Rr]-r[Rr,None]).min(0), (c[Rr]+r[Rr,None]).max(0))
                cst = aL * len(L) + aR * len(Rr)
                if best is None or cst < best[0]: best = (cst, L, Rr)
        rec(best[1]); rec(best[2])
        return A
    root = rec(np.arange(len(c)))
    return inner / root, leafc / root

def main():
    for name, mk in (("random_1024", lambda: scenes.random_spheres(1024)), ("random_256", lambda: scenes.random_spheres(256)), ("mirror_box_256", lambda: scenes.mirror_box(256))):
        sph = api.as_spheres(mk()[0])
        r = np.abs(sph["rad"].astype(np.float64)); c = np.ascontiguousarray(sph["p"]).astype(np.float64)
        med = np.median(r); keep = r <= 16 * med
        c, r = c[keep], r[keep]
        order, code = morton_order(c)
        for label, res in (("morton+count", build_count(order, c, r)), ("median longest axis", build_median(c, r)), ("binned SAH", build_sah(c, r))):
            print(name, label, "inner %.2f  leaf-sphere %.2f  (pair-step cost 50, sphere 20) -> %.0f" % (res[0], res[1], res[0]*50 + res[1]*20))


if __name__ == "__main__":
    main()
