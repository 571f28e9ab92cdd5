"""CPU: the seeded synthetic inputs (openpano_amd/synth.py) are what every timed job and most parity tests run on -- their pixels
are pinned by a checksum, so that a faster generator (round 6: value noise a band of rows at a time) cannot change a workload
unnoticed, and `first=k` cuts exactly the first k views of the full set."""
import zlib

import numpy as np

from openpano_amd import synth


def test_image_set_pixels_are_pinned_and_first_k_is_a_prefix():
    full = synth.image_set(11, 400, 600, seed=22, overlap=0.40)
    assert len(full) == 11 and full[0].shape == (400, 600, 3) and full[0].dtype == np.float32
    assert zlib.crc32(np.stack(full).tobytes()) == CRC_CFG2
    head = synth.image_set(11, 400, 600, seed=22, overlap=0.40, first=3)
    assert len(head) == 3 and all(np.array_equal(a, b) for a, b in zip(head, full))
    # shuffled sets are cut whole (the permutation is over all n views)
    sh = synth.image_set(6, 120, 160, seed=5, overlap=0.45, rows=2, shuffle=True, first=2)
    assert len(sh) == 6


def test_value_noise_bands_equal_whole_planes():
    a = synth._value_noise(np.random.default_rng(7), 150, 333, rows=64)
    b = synth._value_noise(np.random.default_rng(7), 150, 333, rows=1000)       # one band = the whole plane
    c = synth._value_noise(np.random.default_rng(7), 150, 333, rows=7)
    assert np.array_equal(a, b) and np.array_equal(a, c)


CRC_CFG2 = 2586264181            # zlib.crc32 of the 11 stacked views; identical under the generator of rounds 1-5 (checked when the bands went in)
