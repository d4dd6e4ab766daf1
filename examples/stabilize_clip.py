#!/usr/bin/env python3
"""eval.py-style use of the drop-in: stabilise one clip frame by frame (eval.py:93-124), with the
reference's call surface.  The reference reads an .avi with cv2 and loads an .npz checkpoint; neither
ships here, so this example uses a seeded synthetic clip and the seeded synthetic checkpoint.

    python examples/stabilize_clip.py [--frames 8] [--height 288] [--width 512] [--ckpt DIR] [--precision f32|f32x3|f32s|f16]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from coupe.dvsg_amd.clip import stabilize_clip          # noqa: E402
from coupe.dvsg_amd.model import Session, StabNet       # noqa: E402  (reference: `from model import *`)
from coupe.dvsg_amd.weights import make_synthetic_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--height", type=int, default=288)   # config.py:12-13
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--ckpt", default=None, help="reference checkpoint directory (with its `checkpoints` index)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f32x3", "f32s", "f16"],
                    help="f32: float32 matrix instructions (the default); f32x3: float32 tensors and float32-level results from "
                         "three bfloat16 pieces per operand, ~1.3x faster (include/dvsg_amd.h)")
    args = ap.parse_args()
    import inputs
    # stands in for the decoded video: BGR uint8 frames of another size (eval.py:76-81 resizes them)
    frames = (inputs.smooth_frames(1, args.frames, args.height * 3 // 2, args.width * 3 // 2) * 255).astype(np.uint8)
    frames = np.ascontiguousarray(frames[..., ::-1])

    sess = Session()                                              # eval.py:46  tf.Session(...)
    net = StabNet(args.height, args.width)                        # eval.py:50
    net.precision = args.precision
    inputs_, outputs = net.get_evaluation_model(7)
    if args.ckpt:
        net.load_ckpt(args.ckpt, by_score=True)                   # eval.py:56  ckpt_manager.load_ckpt
    else:
        net.load_weights(make_synthetic_weights(seed=0))
    t0 = time.perf_counter()
    stabilised, side_by_side = stabilize_clip(net, sess, frames, side_by_side=True, channel_order="bgr",
                                              as_uint8=True)                       # eval.py:76-124
    dt = time.perf_counter() - t0
    print("stabilised %d frames of %dx%d in %.3f s (%.1f frames/s, autoregressive, batch 1)"
          % (len(stabilised), args.width, args.height, dt, len(stabilised) / dt))
    print("output", stabilised.shape, stabilised.dtype, "side-by-side", side_by_side.shape, side_by_side.dtype)


if __name__ == "__main__":
    main()
