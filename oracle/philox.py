"""NumPy twin of the kernels' Philox4x32-10 draws (csrc/cnerf_dev.hpp).  TEST INFRASTRUCTURE ONLY.

Philox4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11); the reference itself
draws with torch.rand / torch.randn (volumetric_rendering.py:39,106,319), whose CPU stream (MT19937) a device cannot reproduce -- the
in-kernel mode therefore defines its own counter layout: counter = (idx lo, idx hi, stream, offset), key = (seed lo, seed hi);
uniform = top 24 bits of word 0 * 2^-24; normal = Box-Muller on words 0 and 1.  Pinned by the Random123 known-answer vectors
(tests/test_host_cpu.py::test_philox_known_answers)."""
import numpy as np

M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays (or scalars) of counters; returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def _words(seed, offset, stream_id, n):
    idx = np.arange(n, dtype=np.uint64)
    return philox4x32_10((idx & MASK).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32), np.uint32(stream_id), np.uint32(offset),
                         np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))


def uniform(seed, offset, stream_id, n):
    o0, _, _, _ = _words(seed, offset, stream_id, n)
    return ((o0 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def normal(seed, offset, stream_id, n):
    o0, o1, _, _ = _words(seed, offset, stream_id, n)
    u1 = ((o0 >> np.uint32(8)) + np.uint32(1)).astype(np.float32) * np.float32(2.0 ** -24)
    u2 = (o1 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return (np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.2831853071795864) * u2)).astype(np.float32)


def dropout_keep(seed, offset, stream_id, n_points, n_drop, H, p):
    """Keep decisions (n_drop, n_points, H) uint8 of the kernels' dropout under (seed, offset) (include/cnerf.h, cnerf_cfg.drop_p):
    one Philox block per 4 channels, counter index ((point * n_drop + d) * H + c) / 4, word c % 4 keeps iff >= round(p * 2^32).
    stream_id 4 = coarse pass, 5 = fine pass, 6 = cnerf_field_forward."""
    n_blocks = n_points * n_drop * H // 4
    words = np.stack(_words(seed, offset, stream_id, n_blocks), -1)               # (blocks, 4)
    thresh = min(int(p * 4294967296.0 + 0.5), 0xFFFFFFFF)
    keep = (words >= np.uint32(thresh)).astype(np.uint8).reshape(n_points, n_drop, H)
    return np.ascontiguousarray(keep.transpose(1, 0, 2))
