"""TEST INFRASTRUCTURE -- numpy restatement of the library's counter-based unit normals (la_noise_normal_f32, la_misc.hip).

Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) pinned by the known-answer vectors
of the Random123 distribution (tests/test_oracle_l0.py), then the same Box-Muller mapping in float32.  The reference draws its
final-synthesis noise with torch.randn inside G.synthesis (util_latent_aug.py:308): there is no stream to be bit-equal to, the
property that matters is N(0, 1) + reproducibility + independence of the sharding.  Only tests import this file."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over uint32 arrays c0..c3; scalar uint32 key."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0.astype(np.uint64)
        p1 = M1 * c2.astype(np.uint64)
        n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ np.uint32(k0)
        n1 = p1.astype(np.uint32)
        n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ np.uint32(k1)
        n3 = p0.astype(np.uint32)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def noise_normal(rows, row_elems, seed, layer, row0=0):
    """[rows, row_elems] float32: element e of global row row0 + r = Box-Muller of Philox(counter (e // 4, row0 + r, layer, 0), key seed)."""
    q4 = (row_elems + 3) // 4
    q = np.arange(q4, dtype=np.uint64)[None, :].repeat(rows, 0)
    r = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None].repeat(q4, 1)
    x = philox4x32_10(q.astype(np.uint32), r.astype(np.uint32), np.uint32(layer), (q >> np.uint64(32)).astype(np.uint32),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = np.empty([rows, q4, 4], dtype=np.float32)
    for h in range(2):
        u1 = ((x[2 * h] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        u2 = ((x[2 * h + 1] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        ang = (np.float32(6.283185307179586) * u2).astype(np.float32)
        out[:, :, 2 * h] = rad * np.cos(ang)
        out[:, :, 2 * h + 1] = rad * np.sin(ang)
    return out.reshape(rows, q4 * 4)[:, :row_elems]
