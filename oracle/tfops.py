"""Oracle (test infrastructure): float32 restatements of the TF core ops the path uses.

The arithmetic of these ops lives in TensorFlow 1.11 (third-party, not in the reference
tree, not installed here): tf.linspace, tf.matmul on tiny inner dimensions, tf.add_n.
They are restated from the published TF r1.11 CPU kernels; where the summation order
inside a TF kernel is implementation-defined (matmul), the oracle fixes it to the
sequential k order and says so.  parity unpinned (see oracle/__init__.py).
"""
import numpy as np

F32 = np.float32


def tf_linspace(start, stop, num):
    """tf.linspace(start, stop, num) in float32.

    TF r1.11 core/kernels/sequence_ops.cc LinSpaceOp: ``step = (stop - start) / (num - 1)``
    and ``out[i] = start + step * i``, all in T=float (no float64, no exact end point).
    Call sites: ThinPlateSpline.py:94,96; spatial_transformer.py:317-318,474-475.
    """
    num = int(num)
    start = F32(start)
    stop = F32(stop)
    if num == 1:
        return np.array([start], dtype=F32)
    step = F32(F32(stop - start) / F32(num - 1))
    i = np.arange(num, dtype=F32)
    return (start + (step * i).astype(F32)).astype(F32)


def seq_matmul_small(T, G):
    """Batched ``T[B,M,K] @ G[B,K,N]`` (or G[K,N]) in float32, accumulated strictly in
    k = 0..K-1 order with one rounding per multiply and per add (no FMA).

    Stands in for tf.matmul where K is tiny (3, 19, 28): ThinPlateSpline.py:129,163;
    spatial_transformer.py:84,289,438.  Eigen's order is implementation-defined; any order
    is a valid restatement, this one is reproducible on the GPU.
    """
    T = np.asarray(T, dtype=F32)
    G = np.asarray(G, dtype=F32)
    if G.ndim == 2:
        G = G[None]
    K = T.shape[-1]
    acc = (T[:, :, 0:1] * G[:, 0:1, :]).astype(F32)
    for k in range(1, K):
        acc = (acc + (T[:, :, k:k + 1] * G[:, k:k + 1, :]).astype(F32)).astype(F32)
    return acc


def add_n4(a, b, c, d):
    """tf.add_n of four tensors: ((a+b)+c)+d in float32 (ThinPlateSpline.py:89,
    warp_with_optical_flow.py:173, spatial_transformer.py:562)."""
    return (((a + b).astype(F32) + c).astype(F32) + d).astype(F32)
