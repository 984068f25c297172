"""Philox4x32-10 counter-based generator, vectorised with NumPy.

TEST INFRASTRUCTURE ONLY (oracle).  The product's generator is the HIP device
code in radiative-transfer_amd/csrc; this file restates the published
Philox4x32-10 algorithm (Salmon et al., SC'11, "Parallel random numbers: as
easy as 1, 2, 3") so that tests can rebuild the exact uniform streams that
the device kernels consume.

Stream layout shared by every implementation in this repository:

    key     = (seed_lo, seed_hi)
    counter = (photon_id_lo, photon_id_hi, block, atmosphere_index)
    draw j of a photon's walk stream -> block j >> 1, words (2*(j&1), 2*(j&1)+1)
    uniform = ((w_a << 20) | (w_b >> 12)) / 2**52            in [0, 1)

The uniform takes the top 52 bits of the word pair as the fraction of a double in
[1, 2) and subtracts 1: every multiple of 2**-52 in [0, 1) is equally likely.  (The
reference scripts call ``np.random.rand``, MC code.py:496, :479, :436-437, whose
draws are the multiples of 2**-53.)
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)
EMIT_BLOCK = 0x80000000  # blocks >= this are reserved for emission draws


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments broadcastable integer arrays (values < 2**32).

    Returns four uint64 arrays holding the 32-bit output words."""
    c0, c1, c2, c3 = [np.asarray(x, dtype=np.uint64) & MASK32 for x in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def u52(wa, wb):
    """Two 32-bit words -> double in [0,1) with 52 random bits (exact: a 52-bit integer / 2**52)."""
    a = np.asarray(wa, dtype=np.uint64) & MASK32
    b = np.asarray(wb, dtype=np.uint64) & MASK32
    k = (a << np.uint64(20)) | (b >> np.uint64(12))
    return k.astype(np.float64) / 4503599627370496.0


def block_uniforms(photon_id, block, atm, seed):
    """Both uniforms of one Philox block: returns (u_first, u_second)."""
    pid = np.asarray(photon_id, dtype=np.uint64)
    w = philox4x32_10(pid & MASK32, pid >> np.uint64(32), block, atm,
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return u52(w[0], w[1]), u52(w[2], w[3])


def walk_uniform(photon_id, j, atm, seed):
    """Draw j (0-based) of the walk stream of each photon."""
    j = np.asarray(j, dtype=np.uint64)
    ua, ub = block_uniforms(photon_id, j >> np.uint64(1), atm, seed)
    return np.where((j & np.uint64(1)) == 0, ua, ub)


def walk_matrix(n, max_draws, atm, seed, id_base=0):
    """xi[n, max_draws] with xi[i, j] = walk_uniform(id_base + i, j)."""
    ids = (np.arange(n, dtype=np.uint64) + np.uint64(id_base))[:, None]
    blocks = np.arange((max_draws + 1) // 2, dtype=np.uint64)[None, :]
    ua, ub = block_uniforms(ids, blocks, atm, seed)
    out = np.empty((n, 2 * blocks.shape[1]))
    out[:, 0::2] = ua
    out[:, 1::2] = ub
    return np.ascontiguousarray(out[:, :max_draws])


# ---- the opt-in stream with one block per loop iteration and 32-bit uniforms (RT_RNG_PHILOX32X4) ----
def block_words(photon_id, block, atm, seed):
    """The four 32-bit output words of one block."""
    pid = np.asarray(photon_id, dtype=np.uint64)
    return philox4x32_10(pid & MASK32, pid >> np.uint64(32), block, atm, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def u32mid(w):
    """One word -> (w + 1/2) / 2**32 in (0, 1), exact in float64."""
    return (np.asarray(w, dtype=np.uint64) & MASK32).astype(np.float64) * 2.0 ** -32 + 2.0 ** -33


def walk_uniform_packed(photon_id, j, atm, seed):
    """Draw j of a photon's walk in the order the reference consumes them -- step, absorb, azimuth, cosine, step, ...
    -- under the packed layout: block k holds (azimuth, cosine) of the scattering BEFORE flight k in words 0-1 and
    (step, absorb) of flight k in words 2-3."""
    j = np.asarray(j, dtype=np.uint64)
    k, r = j >> np.uint64(2), j & np.uint64(3)
    block = k + (r >= 2).astype(np.uint64)
    w = block_words(photon_id, block, atm, seed)
    word = np.where(r == 0, w[2], np.where(r == 1, w[3], np.where(r == 2, w[0], w[1])))
    return u32mid(word)


def walk_matrix_packed(n, max_draws, atm, seed, id_base=0):
    ids = (np.arange(n, dtype=np.uint64) + np.uint64(id_base))[:, None]
    j = np.arange(max_draws, dtype=np.uint64)[None, :]
    return np.ascontiguousarray(walk_uniform_packed(ids, j, atm, seed))


# ---- the opt-in one-block stream with its 128 bits cut as 24 | 40 | 40 | 24 (RT_RNG_PHILOX40) ----
def walk_uniform_p40(photon_id, j, atm, seed):
    """Draw j of a photon's walk (step, absorb, azimuth, cosine, step, ...) under RT_RNG_PHILOX40: the block and word
    pair of walk_uniform_packed, cut as azimuth 24 bits (top of word 0), cosine 40 (low 8 bits of word 0, word 1),
    step 40 (word 2, top 8 bits of word 3), absorb 24 (low 24 bits of word 3); u = (k + 1/2) / 2**bits."""
    j = np.asarray(j, dtype=np.uint64)
    k, r = j >> np.uint64(2), j & np.uint64(3)
    block = k + (r >= 2).astype(np.uint64)
    w = [x & MASK32 for x in block_words(photon_id, block, atm, seed)]
    step = ((w[2] << np.uint64(8)) | (w[3] >> np.uint64(24))).astype(np.float64)
    absorb = (w[3] & np.uint64(0xFFFFFF)).astype(np.float64)
    azimuth = (w[0] >> np.uint64(8)).astype(np.float64)
    cosine = (((w[0] & np.uint64(0xFF)) << np.uint64(32)) | w[1]).astype(np.float64)
    return np.where(r == 0, (step + 0.5) / 2.0 ** 40, np.where(r == 1, (absorb + 0.5) / 2.0 ** 24,
                    np.where(r == 2, (azimuth + 0.5) / 2.0 ** 24, (cosine + 0.5) / 2.0 ** 40)))


def walk_matrix_p40(n, max_draws, atm, seed, id_base=0):
    ids = (np.arange(n, dtype=np.uint64) + np.uint64(id_base))[:, None]
    j = np.arange(max_draws, dtype=np.uint64)[None, :]
    return np.ascontiguousarray(walk_uniform_p40(ids, j, atm, seed))
