"""CPU: the noise-injection oracle (oracle.add_noise_oracle, restating the reference's add_noise,
graphphysics/dataset/preprocessing.py:177-238, over a counter-based stream).
  * the Philox4x32-10 core against the Random123 known-answer vectors (published with the generator);
  * the reference's own test semantics (tests/graphphysics/dataset/test_preprocessing.py:92-153,235-262):
    NORMAL nodes change, the others do not; t=1 gives no noise, t=0 twenty times the scale; list-valued
    ranges; the two length errors;
  * the draws are standard normal (moments, tails), independent across offsets / ranges."""
import numpy as np
import pytest

from oracle import mgn_oracle as O


def test_philox_known_answer_vectors():
    # Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key) -> output (first two words used here)
    kat = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D)),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E)),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB))]
    for ctr, key, want in kat:
        r0, r1 = O.philox4x32_10(key[0], key[1], *[[c] for c in ctr])
        assert (int(r0[0]), int(r1[0])) == want


def _case(n=4000):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, 5)).astype(np.float32)
    x[:, 3] = rng.choice([0, 0, 0, 4, 5, 6], size=n)   # NORMAL and others
    return x


def test_reference_semantics():
    x = _case()
    normal = x[:, 3] == 0
    y, _ = O.add_noise_oracle(x, 0, 3, 0.1, 3, seed=1)
    assert not np.allclose(y[normal, :3], x[normal, :3]) and np.array_equal(y[~normal], x[~normal])
    assert np.array_equal(y[:, 3:], x[:, 3:])
    y1, _ = O.add_noise_oracle(x, 0, 3, 0.1, 3, t=1, seed=1)      # scale 10*s*(1+cos(pi)) = 0
    assert np.allclose(y1, x, atol=1e-7)
    y0, d = O.add_noise_oracle(x, 0, 3, 0.1, 3, t=0, seed=1)      # scale 20*s
    assert abs((y0 - x)[normal, :3].std() - 2.0) < 0.05
    ym, dm = O.add_noise_oracle(x, [0, 1], [1, 3], [10.0, 20.0], 3, seed=2)
    assert abs((ym - x)[normal, 0].std() - 10) < 0.5 and abs((ym - x)[normal, 1:3].std() - 20) < 0.7
    with pytest.raises(ValueError):
        O.add_noise_oracle(x, [0, 1], [1], 0.1, 3)
    with pytest.raises(ValueError):
        O.add_noise_oracle(x, [0, 1], [1, 3], [0.1], 3)


def test_draws_are_standard_normal_and_independent():
    x = np.zeros((200_000, 4), dtype=np.float32)
    _, d0 = O.add_noise_oracle(x, 0, 3, 1.0, 3, seed=7, offset=0)
    _, d1 = O.add_noise_oracle(x, 0, 3, 1.0, 3, seed=7, offset=1)
    _, d2 = O.add_noise_oracle(x, 0, 3, 1.0, 3, seed=8, offset=0)
    z = d0[0].ravel()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3) < 5e-2
    assert abs((np.abs(z) > 3).mean() - 0.0027) < 5e-4
    for other in (d1[0].ravel(), d2[0].ravel()):
        assert abs(np.corrcoef(z, other)[0, 1]) < 5e-3
    assert abs(np.corrcoef(d0[0][:, 0], d0[0][:, 1])[0, 1]) < 5e-3      # across columns
    _, again = O.add_noise_oracle(x, 0, 3, 1.0, 3, seed=7, offset=0)
    assert np.array_equal(again[0], d0[0])                              # counter-based: reproducible
