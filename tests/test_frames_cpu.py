"""Host logic and oracle known-answer tests for the frame formats either side of the path
(eval.py:76-124): the clip loop's index table, uint8 conversion and the cv2.resize restatement.
cv2 is not installed (parity unpinned, see oracle/frames.py): the resize KATs below follow from
OpenCV's published INTER_LINEAR definition, not from running it."""
import numpy as np
import pytest

from coupe.dvsg_amd.clip import SKIP_LENGTH, teacher_forced_index_table, window_index_table
from oracle import frames as oframes


@pytest.mark.parametrize("n,skip", [(1, SKIP_LENGTH), (2, SKIP_LENGTH), (33, SKIP_LENGTH), (70, SKIP_LENGTH),
                                    (6, (0, 2, 3)), (4, (0,))])
def test_index_table_replays_the_reference_list_manipulation(n, skip):
    trace = oframes.window_index_trace(n, skip)
    want = np.array([[i if kind == 'u' else n + i for kind, i in row] for row in trace])
    got = window_index_table(n, skip)
    assert got.dtype == np.int32 and np.array_equal(got, want)
    # a step never reads a stabilised frame that has not been produced yet
    assert np.all(got[:, :-1] < n + np.arange(n)[:, None]) and np.array_equal(got[:, -1], np.arange(n))


def test_index_table_rejects_offsets_the_reference_loop_cannot_run():
    for bad in [(1, 2, 3), (0, 3, 3), (0, 5, 2), ()]:
        with pytest.raises(ValueError):
            window_index_table(4, bad)


def test_uint8_round_trip_through_the_float_history():
    """eval.py:80 then :112 on an untouched frame gives the frame back, also after the float32
    cast TF applies to what it is fed."""
    u = np.arange(256, dtype=np.uint8)
    assert np.array_equal(oframes.to_uint8(u / 255.), u)
    assert np.array_equal(oframes.to_uint8(np.float32(u / 255.)), u)
    assert np.array_equal(oframes.to_uint8(np.array([0.999999, 0.5, 1e-9])), [254, 127, 0])   # truncation


def test_resize_known_answers():
    rng = np.random.default_rng(5)
    img = rng.uniform(0, 1, (12, 20, 3))
    assert np.array_equal(oframes.resize_linear(img, 20, 12), img)                   # same size: a copy
    const = np.full((9, 7, 3), 0.37)
    assert np.abs(oframes.resize_linear(const, 13, 5) - 0.37).max() < 1e-7          # float32 weights sum to 1
    half = oframes.resize_linear(img, 10, 6)                                         # exact 2x: 2x2 box mean
    box = 0.25 * (img[0::2, 0::2] + img[1::2, 0::2] + img[0::2, 1::2] + img[1::2, 1::2])
    assert np.abs(half - box).max() < 1e-15
    up = oframes.resize_linear(img, 40, 24)                                          # 2x up: edges clamp
    assert np.array_equal(up[0, 0], img[0, 0]) and np.array_equal(up[-1, -1], img[-1, -1])
    assert np.allclose(up[0, 1], 0.75 * img[0, 0] + 0.25 * img[0, 1], atol=1e-7)    # centre (1+.5)/2-.5 = .25
    ramp = np.tile(np.arange(16.0)[None, :, None], (4, 1, 3))                        # linear data stays linear
    r = oframes.resize_linear(ramp, 8, 4)
    assert np.allclose(r[0, :, 0], np.arange(8) * 2 + 0.5, atol=1e-6)


def test_read_frame_flips_bgr_and_scales():
    bgr = np.zeros((4, 6, 3), np.uint8)
    bgr[..., 0], bgr[..., 2] = 255, 51
    f = oframes.read_frame(bgr, 6, 4)
    assert f.dtype == np.float64 and np.array_equal(f[0, 0], [0.2, 0.0, 1.0])


@pytest.mark.parametrize("n,skip", [(33, SKIP_LENGTH), (40, SKIP_LENGTH), (97, SKIP_LENGTH), (6, (0, 2, 3))])
def test_teacher_forced_table_replays_eval_train(n, skip):
    trace = oframes.teacher_forced_trace(n, skip)
    want = np.array([[i if kind == 'u' else n + i for kind, i in row] for row in trace])
    got = teacher_forced_index_table(n, skip)
    assert got.dtype == np.int32 and got.shape == (n - skip[-1], len(skip)) and np.array_equal(got, want)
    assert np.all(got[:, :-1] >= n) and np.all(got[:, -1] < n)     # history: stable clip; current frame: unstable
    with pytest.raises(ValueError):
        teacher_forced_index_table(skip[-1], skip)


def test_uint8_frames_survive_the_division_and_the_scale_exactly():
    """What lets conv1 take uint8 frames (dvsg_stabilize_ring_u8): for EVERY byte value v, the pixel the reference
    feeds is x = float32(v / 255.) (eval.py:80: float64 quotient, cast on feed) and scale_RGB's first op gives
    float32(x * 255) == v exactly (networks.py:8), so the scaled input is float(v) - mean.  And a correctly rounded
    float32 division v / 255 equals the rounded float64 quotient (no double-rounding case among the 256 values): that
    is what the warp's uint8 tap loads compute."""
    v = np.arange(256)
    x = (v / 255.).astype(np.float32)
    assert np.array_equal(x * np.float32(255.0), v.astype(np.float32))
    assert np.array_equal(v.astype(np.float32) / np.float32(255.0), x)


def test_float32_quotient_of_a_byte_by_two_fused_multiply_adds():
    """conv1's masked uint8 staging needs x = float32(v / 255.) itself (eval_train.py:119 `frame / 255.` in float64,
    rounded once by the feed; the mask multiplies x before scale_RGB).  The kernel computes q = v * r,
    x = fma(fma(-q, 255, v), r, q) with r = float32(1 / 255): exact arithmetic on fractions shows it is that value
    for all 256 bytes (a plain v * r is not, for 126 of them)."""
    from fractions import Fraction

    def rn32(fr):   # a Fraction rounded to the nearest float32, ties to even
        f = np.float32(float(fr))
        cands = [np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))]
        return min(cands, key=lambda c: (abs(Fraction(float(c)) - fr), int(np.float32(c).view(np.uint32)) & 1))

    r = np.float32(1.0) / np.float32(255.0)
    plain_bad = 0
    for v in range(256):
        want = np.float32(np.float64(v) / np.float64(255.0))
        q = np.float32(v) * r
        rem = rn32(Fraction(float(v)) - Fraction(float(q)) * 255)
        assert Fraction(float(rem)) == Fraction(float(v)) - Fraction(float(q)) * 255      # the inner fma is exact
        got = rn32(Fraction(float(rem)) * Fraction(float(r)) + Fraction(float(q)))
        assert got == want, v
        assert np.float32(want * np.float32(1.0)) * np.float32(255.0) == np.float32(v)    # mask 1.0: the unmasked staging's float(v)
        plain_bad += q != want
    assert plain_bad > 0
