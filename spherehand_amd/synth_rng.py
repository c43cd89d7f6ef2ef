"""The counter-based random numbers of the synthetic branch, restated in numpy.

HandSynthesizer's kernel path (util_modules.HandSynthesizer: ONE launch, shr_hand_synth_fwd, or three: shr_synth_pose_fwd,
shr_mesh_render_post_fwd, shr_heatmap_render_fwd) draws RandScale's factors, the focal jitter and DepthNoise's per-pixel
shifts and depth noise INSIDE its kernels.  No generator state is carried from draw to draw: a draw is a hash of what it
is for (csrc/common.h rng_hash / rng_key / noise_shift / noise_normal), so the numbers do not depend on the launch
geometry and this module reproduces them on the host -- which is how tests/test_synth_gpu.py checks the noised images
value by value, and how a user regenerates the draws of a call from (seed, call counter).

    seed, counter    HandSynthesizer.rng_state (int64 [4] on the device: seed, call counter, the launch's ticket, unused):
                     seed = torch.initial_seed() when the state was (re)seeded, counter = calls since then; the render
                     launch's LAST workgroup advances it
    key(b, k)        rng_key(seed, counter, sample b, word k): k = 0..2 RandScale, 3 focal jitter, 4 / 5 the sample's
                     pixel-noise keys
    uniform          top 24 bits / 2^24 (torch.rand's float32 construction)
    pixel p          h1 = hash(key4 + p): high half -> x shift, low half -> y shift (16-bit uniforms against the cumulative
                     probabilities of trunc(n sigma + 0.5) in {-1, 0, 1, 2}); h2 = hash(key5 + p): Box-Muller on its
                     halves -> the depth noise of a foreground pixel

The reference draws the same quantities from torch's generators (network/util_modules.py:60-84, :110,
mesh/pointTransformation.py:143-145): parity is in distribution, tested against those formulas.
"""
import math

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def rng_hash(x):
    """lowbias32 on uint32 arrays (csrc/common.h rng_hash)."""
    x = np.asarray(x, dtype=np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def rng_key(seed, counter, b, k):
    """Stream word k of sample b (arrays broadcast) in call `counter` under `seed` (csrc/common.h rng_key)."""
    seed, counter = int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1)
    h = rng_hash((seed & 0xFFFFFFFF) ^ 0x9E3779B9)
    h = rng_hash(h ^ np.uint32(seed >> 32))
    h = rng_hash(h ^ np.uint32(counter & 0xFFFFFFFF))
    h = rng_hash(h ^ np.uint32(counter >> 32) ^ np.uint32(0x85EBCA6B))
    h = rng_hash(h ^ np.asarray(b, np.uint32))
    return rng_hash(h ^ ((np.asarray(k, np.uint64) + np.uint64(0x27D4EB2F)) & _M32).astype(np.uint32))


def uniform(h):
    return (np.asarray(h, np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def sample_draws(seed, counter, B, rand_scale=0.1):
    """draws [6, B] of one call as shr_synth_pose_fwd writes them: s_x, s_y, s_z, focal jitter (float32), and the two
    noise keys (uint32) -- returned as (float32 [4, B], uint32 [2, B])."""
    b = np.arange(B, dtype=np.uint32)
    f = np.empty((4, B), np.float32)
    rs, half = np.float32(rand_scale), np.float32(float(rand_scale) / 2.0)
    for k in range(3):
        f[k] = (uniform(rng_key(seed, counter, b, k)) * rs + np.float32(0.90)) - half
    f[3] = uniform(rng_key(seed, counter, b, 3)) * np.float32(0.2) + np.float32(0.9)
    keys = np.stack([rng_key(seed, counter, b, 4), rng_key(seed, counter, b, 5)])
    return f, keys


def shift_thresholds(sigma_xy=0.5):
    """(P(shift < 0), P(shift < 1), P(shift < 2)) x 65536 of shift = trunc(n sigma + 0.5), n ~ N(0, 1) (truncation
    towards zero, as torch's float -> long cast: shift 0 holds (-1, 1))."""
    Phi = lambda x: 0.5 * math.erfc(-x / math.sqrt(2.0))
    return tuple(int(round(65536.0 * Phi(c / sigma_xy))) for c in (-1.5, 0.5, 1.5))


def noise_field(keys, H, W, sigma_xy=0.5):
    """Per-pixel draws of one call: (dx [B,H,W] int, dy [B,H,W] int, n [B,H,W] float64 standard normal) from the
    samples' keys [2, B] (csrc/common.h noise_shift / noise_normal; the kernels evaluate the normal with the hardware's
    log2 / sqrt / cos: the noised depth agrees to 3e-7, a pixel in millions -- u1 within 2^-16 of 1 -- to 1e-5)."""
    keys = np.asarray(keys, np.uint32)
    B = keys.shape[1]
    p = np.arange(H * W, dtype=np.uint64)[None, :]
    h1 = rng_hash((keys[0].astype(np.uint64)[:, None] + p) & _M32)
    h2 = rng_hash((keys[1].astype(np.uint64)[:, None] + p) & _M32)
    t0, t1, t2 = shift_thresholds(sigma_xy)
    ux, uy = (h1 >> np.uint32(16)).astype(np.int64), (h1 & np.uint32(0xFFFF)).astype(np.int64)
    dx = -1 + (ux >= t0).astype(np.int64) + (ux >= t1) + (ux >= t2)
    dy = -1 + (uy >= t0).astype(np.int64) + (uy >= t1) + (uy >= t2)
    u1 = ((h2 >> np.uint32(16)).astype(np.float64) + 0.5) / 65536.0
    u2 = (h2 & np.uint32(0xFFFF)).astype(np.float64) / 65536.0
    n = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return dx.reshape(B, H, W), dy.reshape(B, H, W), n.reshape(B, H, W)


def depth_noise(scaled, keys, sigma_xy=0.5, sigma_z=0.05):
    """DepthNoise.forward (network/util_modules.py:60-84) on scaled depth [B,H,W] with the draws of noise_field."""
    scaled = np.asarray(scaled, np.float32)
    B, H, W = scaled.shape
    dx, dy, n = noise_field(keys, H, W, sigma_xy)
    v = np.clip(np.arange(H)[None, :, None] + dy, 0, H - 1)
    u = np.clip(np.arange(W)[None, None, :] + dx, 0, W - 1)
    z = scaled[np.arange(B)[:, None, None], v, u]
    noisy = (z + (n.astype(np.float32) * np.float32(sigma_z))).astype(np.float32)
    return np.where(z < np.float32(1.0), noisy, z)
