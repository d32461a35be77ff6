"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by dsf_amd/).

CPU restatement of the reference's evaluation metric ``Trainer.xyz2error`` (train_render.py:826-864) and of the joint
selection of ``Trainer.test_iter`` (:354-383).  Pinned by tests/golden/reference_eval.npz (made by importing the
reference: tests/golden/make_golden_eval.py)."""
import numpy as np

ICVL_BIAS = np.array([20, 22, 13.5, 7.5, 12.5, 12.5, 3, 12.5, 12.5, 8, 16, 12.5, 3, 13, 7.3, 6], dtype=np.float64)


def xyz_to_error(pred, gt, center, cube, dataset="nyu", keep_batch=False, keep_joint=False):
    """pred / gt (B,J,3) cube-normalised, center (B,3) mm, cube (B,3) mm -> mean Euclidean joint error in mm
    (train_render.py:826-851): world = x * cube / 2 + center; icvl subtracts a per-joint depth bias from the
    prediction (:840-842); msra drops joint 0 from the mean (:850-851)."""
    pred, gt = np.asarray(pred), np.asarray(gt)
    B, J, _ = pred.shape
    c = np.tile(np.asarray(center).reshape(B, 1, -1), [1, J, 1])
    s = np.tile(np.asarray(cube).reshape(B, 1, -1), [1, J, 1])
    a = pred * s / 2 + c
    b = gt * s / 2 + c
    if dataset == "icvl":
        a[:, :, 2] = a[:, :, 2] - ICVL_BIAS.reshape(1, 16)
    d = np.sqrt(np.sum((a - b) * (a - b), axis=2))
    if keep_joint:
        return d
    if keep_batch:
        return d.mean(-1)
    return d[:, 1:].mean() if dataset == "msra" else d.mean()


def select_eval_joints(joints, transfer):
    """test_iter (:367-373): reorder by ``mano_layer.transfer`` and drop the last joint of the selection."""
    sel = np.asarray(joints)[:, list(transfer), :]
    return sel[:, :sel.shape[1] - 1, :]
