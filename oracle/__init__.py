"""CPU oracle for the DVSG per-frame inference hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy (float32, op-for-op, in the reference's op order) restatement of
the reference algorithm for the path named by BASELINE.json:north_star:

    scale_RGB -> resnet_v1_50 -> 4 dense layers -> TPS solve -> TPS grid -> sampler A
    (networks.py, model.py, ThinPlateSpline*.py), plus tf_warp (warp_with_optical_flow.py)
    and the affine / projective / elastic spatial transformers (spatial_transformer.py).

Every function cites the reference file:line it follows (paths are into the upstream
reference tree, which is NOT shipped with this repo and is never read at run time).

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker / the CPU number
reported beside the GPU number.  The product (``coupe.dvsg_amd``) never imports it and
has no CPU fallback: without the HIP library it raises.

PARITY STATUS: **parity unpinned**.  The reference is a TensorFlow 1.11 graph program
with no tests, no golden vectors and no fixtures; TensorFlow / tensorlayer are not
installed in the build container (plain ModuleNotFoundError, no network), so neither
the reference nor its third-party arithmetic (tf.contrib.slim resnet_v1_50 @ TF 1.11,
tensorlayer DenseLayer, tf.linspace / tf.matrix_inverse / tf.log kernels) can be run to
produce vectors.  The oracle is pinned only by (a) known-answer tests derived from the
reference source text (tests/test_oracle_kat.py) and
(b) a second, independently written torch-CPU implementation of the CNN
(oracle/cnn_torch.py) that must agree with the NumPy one.
"""
