"""Closed-form synthetic stereo pairs: value-noise texture, constant disparity D.

    mix(a,b): h = a*73856093 ^ b*19349663; h ^= h>>16; h *= 0x85ebca6b; h ^= h>>13;
              h *= 0xc2b2ae35; h ^= h>>16                                   (uint32)
    P_s(x,y) = ((mix(x>>2 + s*4099, y>>2) & 0xFF)*3 + (mix(x + s*4099, y) & 0x3F)) >> 2
    L(x,y) = P_s(x+D, y),  R(x,y) = P_s(x+2D, y)          => true disparity D everywhere
"""
import numpy as np


def _mix(a, b):
    a = a.astype(np.uint32)
    b = b.astype(np.uint32)
    with np.errstate(over="ignore"):
        h = (a * np.uint32(73856093)) ^ (b * np.uint32(19349663))
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return h


def _texture(W, H, s, shift):
    x = np.arange(W, dtype=np.int64)[None, :] + shift
    y = np.arange(H, dtype=np.int64)[:, None]
    coarse = _mix((x >> 2) + s * 4099 + 0 * y, (y >> 2) + 0 * x) & np.uint32(0xFF)
    fine = _mix(x + s * 4099 + 0 * y, y + 0 * x) & np.uint32(0x3F)
    return ((coarse * np.uint32(3) + fine) >> np.uint32(2)).astype(np.uint8)


def synth_pair(W, H, s=0, D=24):
    """Returns (left, right) uint8 [H][W]."""
    return _texture(W, H, s, D), _texture(W, H, s, 2 * D)


def synth_batch(W, H, indices):
    """Pairs of BASELINE.json config 4: pair i uses s = i, D = 8 + (i mod 64)."""
    L = np.empty((len(indices), H, W), np.uint8)
    R = np.empty((len(indices), H, W), np.uint8)
    for j, i in enumerate(indices):
        L[j], R[j] = synth_pair(W, H, int(i), 8 + int(i) % 64)
    return L, R
