"""Synthetic scenes for the BASELINE.json configurations (SURVEY 8d 'Inputs').

Generated with the kernel's own two-stream multiply-with-carry generator (a3) from a fixed
state, in float32, so every run and every machine gets the same bytes."""
import numpy as np

from .api import DIFF, REFR, SPEC, SPHERE_DT
from .host import DEMO_ORIG, DEMO_TARGET, demo_scene


class _Mwc:
    def __init__(self, s0=0x9E3779B9, s1=0x7F4A7C15):
        self.s0, self.s1 = s0, s1

    def next(self):
        self.s0 = (36969 * (self.s0 & 65535) + (self.s0 >> 16)) & 0xFFFFFFFF
        self.s1 = (18000 * (self.s1 & 65535) + (self.s1 >> 16)) & 0xFFFFFFFF
        word = ((self.s0 << 16) + self.s1) & 0xFFFFFFFF
        f = np.array([(word & 0x007FFFFF) | 0x40000000], np.uint32).view(np.float32)[0]
        return np.float32((f - np.float32(2.0)) / np.float32(2.0))

    def uniform(self, lo, hi):
        return np.float32(lo) + np.float32(hi - lo) * self.next()


def _sphere(rad, p, e, c, refl):
    s = np.zeros(1, SPHERE_DT)
    s["rad"], s["p"], s["e"], s["c"], s["refl"] = rad, p, e, c, refl
    return s


def _scatter(rng, count, extent, rad_lo, rad_hi):
    """`count` small spheres resting on the y = 0 plane: 70 % diffuse / 15 % mirror / 15 % glass."""
    out = []
    for _ in range(count):
        rad = rng.uniform(rad_lo, rad_hi)
        px, pz = rng.uniform(-extent, extent), rng.uniform(-extent, extent)
        col = (rng.uniform(.1, .9), rng.uniform(.1, .9), rng.uniform(.1, .9))
        m = rng.next()
        refl = DIFF if m < .7 else (SPEC if m < .85 else REFR)
        out.append(_sphere(rad, (px, rad, pz), (0, 0, 0), col, refl))
    return out


def random_spheres(count=1024):
    """Config C3: ground + one light + (count-2) scattered spheres; Demo camera."""
    rng = _Mwc()
    parts = [_sphere(1000, (0, -1000, 0), (0, 0, 0), (.75, .75, .75), DIFF),
             _sphere(7, (0, 60, 0), (12, 12, 12), (0, 0, 0), DIFF)]
    parts += _scatter(rng, count - 2, 80.0, 1.0, 3.0)
    return np.concatenate(parts), DEMO_ORIG, DEMO_TARGET


def demo_plus(count=16):
    """The north-star '16-sphere scene': the Demo scene + (count-6) scattered spheres."""
    rng = _Mwc()
    parts = [demo_scene()] + _scatter(rng, count - 6, 60.0, 2.0, 6.0)
    return np.concatenate(parts), DEMO_ORIG, DEMO_TARGET


def mirror_box(count=64):
    """Config C5: closed box of six huge diffuse spheres (the classic smallpt box, as the
    reference's cornell.scn uses), one light, (count-7) spheres half mirror / half glass:
    every path runs to the depth limit."""
    rng = _Mwc(0x2545F491, 0x4F6CDD1D)
    w = [
        _sphere(1e4, (1e4 + 1, 40.8, 81.6), (0, 0, 0), (.75, .25, .25), DIFF),
        _sphere(1e4, (-1e4 + 99, 40.8, 81.6), (0, 0, 0), (.25, .25, .75), DIFF),
        _sphere(1e4, (50, 40.8, 1e4), (0, 0, 0), (.75, .75, .75), DIFF),
        _sphere(1e4, (50, 40.8, -1e4 + 270), (0, 0, 0), (.25, .25, .25), DIFF),
        _sphere(1e4, (50, 1e4, 81.6), (0, 0, 0), (.75, .75, .75), DIFF),
        _sphere(1e4, (50, -1e4 + 81.6, 81.6), (0, 0, 0), (.75, .75, .75), DIFF),
        _sphere(7, (50, 66.6, 81.6), (12, 12, 12), (0, 0, 0), DIFF),
    ]
    for i in range(count - 7):
        rad = rng.uniform(3.0, 7.0)
        p = (rng.uniform(10, 90), rng.uniform(rad, 60), rng.uniform(20, 140))
        w.append(_sphere(rad, p, (0, 0, 0), (.9, .9, .9), SPEC if i % 2 == 0 else REFR))
    return np.concatenate(w), (50.0, 45.0, 205.6), (50.0, 44.957388, 204.6)
