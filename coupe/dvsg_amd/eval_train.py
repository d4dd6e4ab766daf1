"""Drop-in for the evaluation graph of the reference's eval_train.py on MI355X.

eval_train.py does not use `StabNet.get_evaluation_model`: it builds its own graph
(eval_train.py:25-51) in which the regressor's input is `patches_masked_t = patches_t * mask`
(:43-45), `random_mask` (:53-64) being model.py:156-167 -- an all-ones image of the 18 history
channels warped by `ProjectiveTransformer` with a near-identity random homography; the TPS warps
still sample the UNMASKED `u_t` (:48-49).  Same call surface:

    inputs, outputs = get_evaluation_model(sample_num, param_dim, num_control_points, h, w)
    model_of(outputs).load_weights(npz_dict)            # stands in for ckpt_manager.load_ckpt(sess)
    s_t_pred = Session().run(outputs['s_t_pred'], {inputs['patches_t']: x, inputs['u_t']: x[..., 18:]})

The graph's only stochastic op, `tf.random_uniform` (:55), is drawn with `torch.rand` per `run`
(`model_of(outputs).mask_generator` seeds it) unless the OPTIONAL extra feed `inputs['random_H']`
[B,8] -- H after the scale / identity offset of :56-57 -- supplies it; that is how the parity tests
pin the graph against the oracle.  Compute: `dvsg_random_mask_plane_f32` (one [B,H,W] plane) and
`dvsg_stabilize_masked_f32`, which multiplies the plane into the history channels inside conv1's load
stage; `patches_masked_t` / `random_masks_t` exist as tensors only when they are fetched.
The clip loop of eval_train.py:137-165 is `coupe.dvsg_amd.clip.stabilize_clip_teacher_forced`.
"""
from .model import Session, StabNet  # noqa: F401


def get_evaluation_model(sample_num, param_dim, num_control_points, h, w):
    """eval_train.py:25-51.  Returns (inputs, outputs) OrderedDicts with the reference's keys
    (`patches_t`, `u_t` | `V_src`, `patches_masked_t`, `random_masks_t`, `F_t`, `s_t_pred`, `x_offset_t`,
    `y_offset_t`, `s_t_pred_mask`) plus the optional feed `inputs['random_H']`."""
    if num_control_points != 5 or param_dim != num_control_points ** 2:
        raise ValueError("the reference's dense4 has 50 outputs: num_control_points must be 5 and param_dim 25 "
                         "(eval_train.py:84-85)")
    net = StabNet(h, w)
    net.masked = True
    return net.get_evaluation_model(sample_num)


def model_of(outputs):
    """The `StabNet` behind a graph returned by `get_evaluation_model` (to load weights, pick a precision,
    seed `mask_generator`)."""
    return outputs['F_t'].model
