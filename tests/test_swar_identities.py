"""The byte arithmetic k_hash.h builds from 16-bit packed operations, checked exhaustively in numpy (no GPU): these are the
identities its comments claim, for every byte pair and every tau.  Reference semantics: _mm_subs_epi8 (filter.hpp:649-651)."""
import numpy as np


def sat16(x):
    return np.clip(x, -32768, 32767)


def lanes(hi, lo):
    """signed 16-bit lane with `hi` in the high byte (as int8) and `lo` (0..255) below it"""
    return hi.astype(np.int64) * 256 + lo.astype(np.int64)


def high_byte(lane):
    return (lane >> 8).astype(np.int64)        # arithmetic shift: the int8 in the high byte


S = np.arange(-128, 128, dtype=np.int64)[:, None]     # the int8 operand
G = np.arange(0, 256, dtype=np.int64)[None, :]        # whatever byte lies below it in the lane


def test_saturating_subtract_in_the_high_byte_ignores_the_low_byte():
    for tau in range(-128, 128):
        want = np.clip(S - tau, -128, 127) + 0 * G
        got = high_byte(sat16(lanes(S, G) - tau * 256))
        assert np.array_equal(got, want), tau


def test_complemented_saturating_subtract_with_the_constant_as_minuend():
    """~clamp(s - tau) = high byte of sat16(((tau - 1) * 256 + 255) - lane) for -127 <= tau <= 127, whatever the low byte (the
    form k_hash used first: tau = -128 has no such minuend in 16 bits)."""
    for tau in range(-127, 128):
        want = ~np.clip(S - tau, -128, 127) + 0 * G
        minuend = (tau - 1) * 256 + 255
        assert -32768 <= minuend <= 32767
        got = high_byte(sat16(minuend - lanes(S, G)))
        assert np.array_equal(got, want), tau


def test_complemented_saturating_subtract_every_tau_with_the_low_byte_forced():
    """subs_epi8x4_not: with the byte below the int8 forced to 255, ~clamp(s - tau) = high byte of sat16(tau * 256 - lane) for
    EVERY tau in int8 -- the minuend is subs_epi8x4's own constant."""
    for tau in range(-128, 128):
        want = ~np.clip(S - tau, -128, 127)
        assert -32768 <= tau * 256 <= 32767
        got = high_byte(sat16(tau * 256 - lanes(S, np.full_like(S, 255))))
        assert np.array_equal(got, want), tau


def test_unsigned_saturating_add_in_the_high_byte_ignores_the_low_byte():
    X = np.arange(0, 256, dtype=np.int64)[:, None]
    for t in range(0, 256):
        want = np.minimum(X + t, 255) + 0 * G
        got = np.minimum(X * 256 + G + t * 256, 65535) >> 8
        assert np.array_equal(got, want), t


def test_lerp_compare_polarities():
    """v_lerp_u8(x, y, c) = (x + y + c) >> 1 per byte on a 9-bit sum.  (b, ~a, 1): bit 7 = b >= a.  (a, ~b, 0): bit 7 = a > b."""
    a = np.arange(256, dtype=np.int64)[:, None]
    b = np.arange(256, dtype=np.int64)[None, :]
    assert np.array_equal((((b + (255 - a) + 1) >> 1) >> 7) & 1, (b >= a).astype(np.int64))
    assert np.array_equal((((a + (255 - b) + 0) >> 1) >> 7) & 1, (a > b).astype(np.int64))


def test_bitop3_tables():
    """v_bitop3_b32 table index = a * 4 + b * 2 + c.  The plane inserts -- 0xE4: c ? a : b; 0x4E: c ? ~a : b -- and the
    store phase of the batched SSE instantiations (k_hash.h, INV: a = the complemented code, b = "row is hashed", c = the
    candidate mask / the running OR): 0xAE: c | (~a & b); 0x5D: c ? (~a & b) : 1; 0xD5 (a = the plain code): c ? (a & b) : 1."""
    tables = (
        (0xE4, lambda a, b, c: a if c else b),
        (0x4E, lambda a, b, c: (1 - a) if c else b),
        (0xAE, lambda a, b, c: c | ((1 - a) & b)),
        (0x5D, lambda a, b, c: ((1 - a) & b) if c else 1),
        (0xD5, lambda a, b, c: (a & b) if c else 1),
    )
    for tt, f in tables:
        for a in (0, 1):
            for b in (0, 1):
                for c in (0, 1):
                    assert (tt >> (a * 4 + b * 2 + c)) & 1 == f(a, b, c), hex(tt)


def test_complemented_plane_forms_of_the_transpose():
    """k_hash keeps the planes as NOT(code bit) and, in the batched SSE instantiations (INV), the codes complemented through
    the byte transposes.  The complemented forms of the two planes that are not plain complements:
        q0:  ~(~p0 | ((~p8 >> 7) & m8))  ==  p0 & ((p8 >> 7) | ~m8)           (test 8 OR-ed into bit 0 where m8 selects)
        q3:  ~((~p3 >> s) & m3)          ==  (p3 >> s) | ~m3                   (the last plane's n3 = 8 - s tests)
    for every 32-bit word, both m8 values and every n3 (numpy, random words + the corner words)."""
    rng = np.random.default_rng(5)
    words = np.concatenate([rng.integers(0, 1 << 32, 20000, dtype=np.uint64), np.array([0, 0xFFFFFFFF, 0x80808080, 0x7F7F7F7F, 0x01010101], np.uint64)])
    M = np.uint64(0xFFFFFFFF)
    p0, p8, p3 = words, np.roll(words, 7), np.roll(words, 13)
    for m8 in (np.uint64(0x01010101), np.uint64(0x01010100)):
        plain = (~p0 & M) | (((~p8 & M) >> np.uint64(7)) & m8)
        inv = p0 & ((p8 >> np.uint64(7)) | (~m8 & M))
        assert np.array_equal(~plain & M, inv)
    for n3 in range(1, 8):
        m3 = np.uint64(0x01010101 * ((1 << n3) - 1))
        s = np.uint64(8 - n3)
        plain = ((~p3 & M) >> s) & m3
        inv = (p3 >> s) | (~m3 & M)
        # the shift drags bits of the neighbouring byte in; only the bits m3 selects are code bits -- the others are
        # all-ones in the complemented form (code bit 0) and zero in the plain one
        assert np.array_equal(~plain & M, inv)


def test_first_test_of_a_plane_may_carry_garbage_below_bit_7():
    """k_hash's planes take the first compare word of a plane as it is (bit 7 of each byte = the test, bits 0 .. 6 garbage)
    instead of inserting it into the plane's initial value.  After n inserts in all (each: plane = (plane >> 1) with bit 7 of
    every byte replaced) the words read from the planes are the same as with a clean start: the full planes (n = 8) bit for
    bit, P8 (n = 1) through (p8 >> 7) under m8, the last plane (n = n3 < 8) through the shift by 8 - n3 under m3."""
    rng = np.random.default_rng(11)
    M, Hh = np.uint64(0xFFFFFFFF), np.uint64(0x80808080)
    K = 5000
    for n in range(1, 9):
        ge = rng.integers(0, 1 << 32, (n, K), dtype=np.uint64)          # compare words: only bit 7 of each byte means anything
        clean = np.full(K, 0xFFFFFFFF, np.uint64)                        # the initial value of the planes ("no bit")
        short = None
        for i in range(n):
            clean = ((clean >> np.uint64(1)) & ~Hh & M) | (ge[i] & Hh)
            short = ge[i] if i == 0 else (((short >> np.uint64(1)) & ~Hh & M) | (ge[i] & Hh))
        if n == 8:
            assert np.array_equal(clean, short)
        if n == 1:
            for m8 in (np.uint64(0x01010101), np.uint64(0x01010100)):
                assert np.array_equal((clean >> np.uint64(7)) | (~m8 & M), (short >> np.uint64(7)) | (~m8 & M))
                assert np.array_equal(((~clean & M) >> np.uint64(7)) & m8, ((~short & M) >> np.uint64(7)) & m8)
        if n < 8:
            m3 = np.uint64(0x01010101 * ((1 << n) - 1))
            s = np.uint64(8 - n)
            assert np.array_equal((clean >> s) | (~m3 & M), (short >> s) | (~m3 & M))
            assert np.array_equal(((~clean & M) >> s) & m3, ((~short & M) >> s) & m3)
