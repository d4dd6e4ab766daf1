"""TensorFlow checkpoint reader without TensorFlow (coupe/dvsg_amd/tf_checkpoint.py; SURVEY.md 8f-4, model.py:125-154).
PARITY UNPINNED: TensorFlow is not installed and the reference ships no checkpoint, so these are round trips through
the module's own writer plus known answers of the published formats (CRC-32C check value, a hand-assembled Snappy
stream, hand-assembled protobuf bytes)."""
import struct

import numpy as np
import pytest

from coupe.dvsg_amd import tf_checkpoint as tfc
from coupe.dvsg_amd.weights import PREFIX, make_synthetic_weights, validate


def test_crc32c_and_mask_known_answers():
    assert tfc.crc32c(b"123456789") == 0xE3069283            # the CRC-32C check value (RFC 3720 appendix B.4)
    assert tfc.crc32c(b"") == 0
    c = tfc.crc32c(b"123456789")
    assert tfc.masked_crc(b"123456789") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def test_snappy_literals_and_overlapping_copies():
    # "abcdabcdabcdabcdXYZ": literal "abcd", copy len 12 offset 4 (overlaps its own output), literal "XYZ"
    stream = bytes([19]) + bytes([3 << 2]) + b"abcd" + bytes([((12 - 1) << 2) | 2]) + struct.pack("<H", 4) + bytes([2 << 2]) + b"XYZ"
    assert tfc.snappy_decompress(stream) == b"abcdabcdabcdabcdXYZ"
    # a long literal (length in a trailing byte) and a 1-byte-offset copy (3-bit length field, 11-bit offset)
    lit = bytes(range(70))
    stream = bytes([80]) + bytes([60 << 2, 69]) + lit + bytes([((10 - 4) << 2) | 1, 70])
    assert tfc.snappy_decompress(stream) == lit + lit[:10]
    with pytest.raises(ValueError):
        tfc.snappy_decompress(bytes([5]) + bytes([((4 - 4) << 2) | 1, 9]))      # copy from before the start


def test_table_round_trip_with_prefix_compression_and_many_blocks(tmp_path):
    items = [(b"", b"header")] + [(("resnet_v1_50/block%d/unit_%d/bottleneck_v1/conv%d/weights" % (b, u, c)).encode(),
                                   bytes([b, u, c]) * (7 * u)) for b in range(1, 5) for u in range(1, 7) for c in range(1, 4)]
    path = str(tmp_path / "t.index")
    tfc.write_table(path, items, block_size=256, restart_interval=4)
    assert tfc.read_table(path) == sorted(items)
    data = bytearray(open(path, "rb").read())
    data[10] ^= 0xFF                                         # a flipped byte inside the first block
    open(path, "wb").write(bytes(data))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read_table(path)
    assert tfc.read_table(path, verify=False)[0] == (b"", b"header")       # unchecked: whatever the bytes say
    open(path, "wb").write(b"not a table")
    with pytest.raises(ValueError, match="magic"):
        tfc.read_table(path)


def test_bundle_and_v1_round_trips(tmp_path):
    rng = np.random.default_rng(0)
    arrays = {"a/weights": rng.standard_normal((3, 3, 8, 16)).astype(np.float32),
              "a/BatchNorm/gamma": rng.standard_normal(16).astype(np.float32),
              "global_step": np.array(1234567, dtype=np.int64),
              "half": rng.standard_normal((5, 2)).astype(np.float16),
              "dbl": rng.standard_normal(4)}
    v2 = str(tmp_path / "model.ckpt")
    tfc.write_bundle(v2, arrays)
    v1 = str(tmp_path / "old.ckpt")
    tfc.write_v1(v1, arrays)
    for path in (v2, v1):
        got = tfc.load_checkpoint(path)
        assert sorted(got) == sorted(arrays)
        for k in arrays:
            assert got[k].dtype == arrays[k].dtype and got[k].shape == arrays[k].shape and np.array_equal(got[k], arrays[k])
    raw = bytearray(open(v2 + ".data-00000-of-00001", "rb").read())
    raw[3] ^= 1
    open(v2 + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="data checksum"):
        tfc.load_checkpoint(v2)
    with pytest.raises(FileNotFoundError):
        tfc.load_checkpoint(str(tmp_path / "missing.ckpt"))


def test_v1_tensor_proto_with_packed_float_val(tmp_path):
    """TF's V1 saver stores floats in TensorProto.float_val (packed), not tensor_content: assembled by hand here."""
    vals = np.array([1.5, -2.25, 3.0, 0.125, 7.0, -8.5], dtype=np.float32)
    shape = b"\x12\x02\x08\x02" + b"\x12\x02\x08\x03"                       # dims 2, 3
    tp = b"\x08\x01" + b"\x12" + bytes([len(shape)]) + shape + b"\x2a" + bytes([vals.nbytes]) + vals.tobytes()
    name = b"conv/weights"
    ss = b"\x0a" + bytes([len(name)]) + name + b"\x12\x04\x0a\x00\x0a\x00" + b"\x1a" + bytes([len(tp)]) + tp
    path = str(tmp_path / "v1.ckpt")
    tfc.write_table(path, [(b"", b"\x0a\x00"), (b"\x00conv/weights\x00\x01", b"\x12" + bytes([len(ss)]) + ss)])
    got = tfc.load_checkpoint(path)
    assert list(got) == ["conv/weights"] and np.array_equal(got["conv/weights"], vals.reshape(2, 3))


def test_init_from_slim_checkpoint_restores_the_trunk_but_not_conv1(tmp_path):
    """model.py:125-154: every slim model variable of localizationNet except resnet_v1_50/conv1/* comes from the ImageNet
    checkpoint (whose root conv has 3 input channels); the dense head is not a slim variable and stays."""
    base = make_synthetic_weights(seed=0)
    donor = make_synthetic_weights(seed=1)
    ckpt = {}
    for k, v in donor.items():
        name = k[:-2]
        if name.startswith(PREFIX + "resnet_v1_50/"):
            ckpt[name[len(PREFIX):]] = v
    ckpt["resnet_v1_50/conv1/weights"] = np.zeros((7, 7, 3, 64), np.float32)      # ImageNet root: RGB input
    ckpt["resnet_v1_50/logits/weights"] = np.zeros((1, 1, 2048, 1000), np.float32)
    path = str(tmp_path / "resnet_v1_50.ckpt")
    tfc.write_bundle(path, ckpt)
    merged = tfc.init_from_slim_checkpoint(base, path)
    validate(merged)
    n_trunk = 0
    for k in base:
        name = k[:-2]
        if name.startswith(PREFIX + "resnet_v1_50/conv1/") or "/df/" in name:
            assert np.array_equal(merged[k], base[k]), k
        else:
            assert np.array_equal(merged[k], donor[k]), k
            n_trunk += 1
    assert n_trunk == 52 * 5
    del ckpt["resnet_v1_50/block3/unit_2/bottleneck_v1/conv2/BatchNorm/moving_variance"]
    tfc.write_bundle(path, ckpt)
    with pytest.raises(KeyError, match="moving_variance"):
        tfc.init_from_slim_checkpoint(base, path)
    ckpt["resnet_v1_50/block3/unit_2/bottleneck_v1/conv2/BatchNorm/moving_variance"] = np.zeros(7, np.float32)
    tfc.write_bundle(path, ckpt)
    with pytest.raises(ValueError, match="shape"):
        tfc.init_from_slim_checkpoint(base, path)
