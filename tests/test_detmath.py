"""The restated sinf/cosf/powf (oracle om_*) against the host libm, exhaustively over every
argument the renderer can produce.  Meaningful where the host libm is glibc 2.35 with the FMA
variants (this image); elsewhere the comparison is skipped, the HIP-vs-oracle tests still hold."""
import platform
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import _oracle as O


def _glibc_235():
    return platform.libc_ver() == ("glibc", "2.35")


@pytest.mark.skipif(not _glibc_235(), reason="restates glibc 2.35's libm")
def test_sincos_all_renderer_arguments():
    # x = (2*pi)_f32 * k/2^23 for every 23-bit k: every value GetRandom can return
    assert O.oracle().orc_math_mismatches(0, 0, 1 << 23) == 0


@pytest.mark.skipif(not _glibc_235(), reason="restates glibc 2.35's libm")
def test_gamma_pow_every_float_in_unit_interval():
    n = 0x3F800001            # bit patterns of [0, 1]
    parts = 32
    step = (n + parts - 1) // parts
    with ThreadPoolExecutor(8) as ex:
        bad = list(ex.map(lambda i: O.oracle().orc_math_mismatches(1, i * step, min(n, (i + 1) * step)),
                          range(parts)))
    assert sum(bad) == 0


def test_special_values():
    lib = O.oracle()
    assert lib.om_gammaf(0.0) == 0.0
    assert lib.om_gammaf(1.0) == 1.0
    assert lib.om_sinf(0.0) == 0.0
    assert lib.om_cosf(0.0) == 1.0
    assert lib.orc_to_int(0.0) == 0
    assert lib.orc_to_int(1.0) == 255
    assert lib.orc_to_int(7.5) == 255
    assert lib.orc_to_int(-3.0) == 0
    assert lib.orc_to_int(float("nan")) == 0


def test_rng_post_operations_fold_into_one_exact_operation():
    """The kernel forms GetRandom() = (f - 2) / 2, GetRandom() - 0.5 and 1 - 2 * GetRandom()
    (RayTracing_Kernel.cl:143-169, :507-508, :204) as fma(f, .5, -1), fma(f, .5, -1.5) and 3 - f.
    All 2^23 values of f: the folded forms are the same binary32 numbers (every step is exact)."""
    k = np.arange(1 << 23, dtype=np.uint32)
    f = (k | np.uint32(0x40000000)).view(np.float32)
    u = (f - np.float32(2)) / np.float32(2)
    fma = lambda c: (f.astype(np.float64) * 0.5 + c).astype(np.float32)      # exact in binary64: one rounding
    assert np.array_equal(fma(-1.0).view(np.uint32), u.view(np.uint32))
    assert np.array_equal(fma(-1.5).view(np.uint32), (u - np.float32(0.5)).view(np.uint32))
    assert np.array_equal((np.float32(3) - f).view(np.uint32), (np.float32(1) - np.float32(2) * u).view(np.uint32))
