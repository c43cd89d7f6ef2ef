"""The numpy restatement of the synthesizer's counter-based generator (spherehand_amd/synth_rng.py): known answers of the
hash, the shift table against the normal CDF, and the draws' ranges -- no GPU."""
import math

import numpy as np


def test_lowbias32_known_answers_and_bijectivity():
    from spherehand_amd import synth_rng
    # lowbias32 by hand for x = 1: every step written out
    x = 1
    x ^= x >> 16; x = (x * 0x7FEB352D) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846CA68B) & 0xFFFFFFFF; x ^= x >> 16
    assert int(synth_rng.rng_hash(np.uint32(1))) == x
    assert int(synth_rng.rng_hash(np.uint32(0))) == 0
    h = synth_rng.rng_hash(np.arange(1 << 20, dtype=np.uint32))
    assert len(np.unique(h)) == 1 << 20                       # (an invertible mix: no collisions)
    bits = np.unpackbits(h.view(np.uint8)).mean()
    assert abs(bits - 0.5) < 1e-3


def test_shift_table_is_the_truncated_normal():
    from spherehand_amd import synth_rng
    t0, t1, t2 = synth_rng.shift_thresholds(0.5)
    Phi = lambda v: 0.5 * math.erfc(-v / math.sqrt(2.0))
    assert (t0, t1, t2) == (round(65536 * Phi(-3.0)), round(65536 * Phi(1.0)), round(65536 * Phi(3.0)))
    # the mass beyond shifts -1 .. +2 at the reference's sigma
    assert Phi(-5.0) + (1.0 - Phi(5.0)) < 1e-6


def test_draws_are_deterministic_and_in_range():
    from spherehand_amd import synth_rng
    f1, k1 = synth_rng.sample_draws(5, 3, 1000)
    f2, k2 = synth_rng.sample_draws(5, 3, 1000)
    f3, k3 = synth_rng.sample_draws(5, 4, 1000)
    assert np.array_equal(f1, f2) and np.array_equal(k1, k2) and not np.array_equal(f1, f3) and not np.array_equal(k1, k3)
    assert (f1[:3] >= 0.85 - 1e-6).all() and (f1[:3] <= 0.95 + 1e-6).all() and (f1[3] >= 0.9).all() and (f1[3] <= 1.1 + 1e-6).all()
    dx, dy, n = synth_rng.noise_field(k1[:, :4], 32, 32)
    assert dx.min() >= -1 and dx.max() <= 2 and abs(n.mean()) < 0.1 and abs(n.std() - 1.0) < 0.1
    z = np.full((4, 32, 32), 1.0, np.float32); z[:, 8:24, 8:24] = 0.5
    out = synth_rng.depth_noise(z, k1[:, :4])
    assert out.dtype == np.float32 and (out[out >= 1.0] == 1.0).all() and (np.abs(out[out < 1.0] - 0.5) < 0.4).all()
