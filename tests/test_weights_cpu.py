"""Checkpoint plumbing (ckpt_manager.py:15-33,42; SURVEY.md 8b 'Weights / on-disk')."""
import os

import numpy as np
import pytest

from coupe.dvsg_amd import weights as W


def test_synthetic_checkpoint_has_the_reference_inventory(synthetic_weights):
    assert len(synthetic_weights) == 273
    n_params = sum(v.size for v in synthetic_weights.values())
    assert 30.0e6 < n_params < 31.0e6
    k = "stabNet/localizationNet/resnet_v1_50/conv1/weights:0"
    assert synthetic_weights[k].shape == (7, 7, 21, 64)
    assert synthetic_weights["stabNet/localizationNet/df/dense4/W:0"].shape == (512, 50)
    assert len(W.conv_specs()) == 53
    again = W.make_synthetic_weights(seed=0)
    assert all(np.array_equal(again[k], v) for k, v in synthetic_weights.items())


def test_validate_accepts_both_name_styles_and_rejects_damage(synthetic_weights):
    W.validate(synthetic_weights)
    W.validate({k[:-2]: v for k, v in synthetic_weights.items()})          # without ':0'
    extra = dict(synthetic_weights)
    extra["global_step:0"] = np.zeros(1, np.float32)                       # unknown keys ignored
    W.validate(extra)
    bad = dict(synthetic_weights)
    del bad["stabNet/localizationNet/df/dense2/b:0"]
    with pytest.raises(ValueError, match="missing"):
        W.validate(bad)
    bad = dict(synthetic_weights)
    bad["stabNet/localizationNet/resnet_v1_50/block1/unit_1/bottleneck_v1/conv2/weights:0"] = \
        np.zeros((3, 3, 64, 32), np.float32)
    with pytest.raises(ValueError, match="shape"):
        W.validate(bad)


def test_ckpt_dir_index_semantics(tmp_path, synthetic_weights):
    small = {k: v for k, v in list(synthetic_weights.items())[:5]}
    np.savez(tmp_path / "DVSG_00003.npz", **small)
    np.savez(tmp_path / "DVSG_00007.npz", **{k: v + 1 for k, v in small.items()})
    # sorted by score, most recent duplicated on the last line (ckpt_manager.py:58-62,66-88)
    (tmp_path / "checkpoints").write_text("DVSG_00003.npz 0.11\nDVSG_00007.npz 0.25\nDVSG_00007.npz 0.25\n")
    best = W.load_ckpt_dir(str(tmp_path), by_score=True)
    last = W.load_ckpt_dir(str(tmp_path), by_score=False)
    k = next(iter(small))
    assert np.array_equal(best[k], small[k]) and np.array_equal(last[k], small[k] + 1)
    with pytest.raises(FileNotFoundError):
        W.load_ckpt_dir(str(tmp_path / "nowhere"))
