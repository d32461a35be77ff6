"""ctypes binding of libdsf_hip.so (the C ABI declared in include/dsf_hip.h).

There is NO CPU fallback: if the shared library is missing, or an op is handed a
non-GPU tensor, the call raises.  The library is built in-tree by
``dsf_amd/csrc/build.sh`` (``__graft_entry__.build()``).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdsf_hip.so")
ERR_UNSUPPORTED = 2         # DSF_ERR_UNSUPPORTED (include/dsf_hip.h): the launcher declined the shape / mode, nothing was launched
EXPECTED_ABI = 5            # dsf_abi_version() of the library these bindings were written for (csrc/api.hip)

c_float_p = ctypes.c_void_p
_lib = None
WRITE_EPOCH = [0]      # bumped by nn_conv.weights_changed(): parameter writes torch's version counters cannot see


class MissingNativeLibrary(RuntimeError):
    pass


class dsf_mano_model(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "v_template", "shapedirs", "posedirs", "j_regressor", "j_template", "j_shapedirs", "hands_comp",
        "hands_mean", "weights", "parents", "wrist_ring", "jreg_rowptr", "jreg_col", "jreg_val")]


class dsf_camera(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("fx", "fy", "px", "py", "img_w", "img_h")]


class dsf_sphere_model(ctypes.Structure):
    _fields_ = [("jreg_mask", ctypes.c_void_p), ("coll_mask", ctypes.c_void_p),
                ("t_finger", ctypes.c_float * 3), ("t_palm", ctypes.c_float * 4)]


# every symbol include/dsf_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "dsf_abi_version", "dsf_status_string", "dsf_mano_forward", "dsf_mano_backward", "dsf_project_face_verts",
    "dsf_rasterize_meshes", "dsf_rasterize_meshes_backward", "dsf_crop_setup", "dsf_render_crop_forward",
    "dsf_render_crop_backward", "dsf_point_face_dist_forward", "dsf_point_face_dist_backward",
    "dsf_mesh_point_dist_forward", "dsf_mesh_point_dist_backward", "dsf_sphere_set", "dsf_collision_forward",
    "dsf_collision_backward", "dsf_seg_pcl", "dsf_uvd_to_xyz", "dsf_xyz_to_uvd", "dsf_uvd_to_xyz_backward",
    "dsf_xyz_to_uvd_backward", "dsf_crop_hand", "dsf_img2pcl", "dsf_joint2offset_forward",
    "dsf_joint2offset_backward", "dsf_offset2joint_forward", "dsf_offset2joint_backward",
    "dsf_conv_igemm_forward", "dsf_conv_igemm_bwd_data_s1", "dsf_conv_igemm_forward_wt", "dsf_conv_igemm_wrw",
    "dsf_depth_crop_normalize", "dsf_depth_crop_normalize_u16", "dsf_conv_c1_supported", "dsf_conv_c1_workspace_bytes", "dsf_conv_c1_forward", "dsf_conv_c1_wrw",
    "dsf_conv_x6_image_bytes", "dsf_conv_x6_split_weights", "dsf_conv_x6_image_granules",
    "dsf_conv_x6_split_weights_multi", "dsf_conv_x6_forward", "dsf_conv_x6_wrw", "dsf_mfma_bf16_probe", "dsf_bn_forward", "dsf_bn_apply", "dsf_bn_backward", "dsf_bn_workspace_bytes", "dsf_col_sum", "dsf_col_sum_workspace_bytes",
    "dsf_huber_mean_forward", "dsf_huber_mean_backward", "dsf_adamw_multi", "dsf_adamw_chunk_elems",
    "dsf_maxpool_forward", "dsf_maxpool_backward", "dsf_part_volume_workspace_bytes", "dsf_part_intersection_volume",
    "dsf_set_deterministic", "dsf_get_deterministic", "dsf_conv_x6_wrw_workspace_bytes", "dsf_conv_x6_wrw_ws",
    "dsf_depth_augment_crop", "dsf_conv_x6_bn_stats_rows", "dsf_conv_x6_forward_bn", "dsf_conv_x6_forward_affine", "dsf_bn_forward_from_stats",
    "dsf_instnorm_forward", "dsf_reflect_pad_nhwc", "dsf_bn_local_sums", "dsf_bn_forward_from_sums", "dsf_bn_backward_sums", "dsf_bn_backward_apply",
    "dsf_bn_acc_rows", "dsf_conv_x6_forward_splits", "dsf_conv_x6_forward_into", "dsf_conv_x6_forward_bn_acc", "dsf_bn_forward_acc", "dsf_bn_backward_acc",
    "dsf_conv_x6_forward_plan", "dsf_conv_co1_forward", "dsf_conv_x6_wrw_bias", "dsf_bn_backward_pair", "dsf_bn_backward_acc_pair",
    "dsf_m2d_forward", "dsf_m2d_backward", "dsf_cube_points_forward", "dsf_cube_points_backward", "dsf_view_rotate", "dsf_part_mean_forward",
    "dsf_part_mean_backward", "dsf_mano_reg_forward", "dsf_mano_reg_backward", "dsf_cube_normalise", "dsf_m2p_forward", "dsf_m2p_backward", "dsf_sphere_mixed", "dsf_offset2joint_forward_strided", "dsf_offset2joint_backward_strided", "dsf_pool_linear_forward", "dsf_pool_linear_backward",
    "dsf_offset2joint_cl_workspace_floats", "dsf_offset2joint_forward_cl", "dsf_offset2joint_backward_cl",
    "dsf_bn_relu_pool_forward", "dsf_bn_relu_pool_backward", "dsf_conv_c1_forward_bn_acc", "dsf_conv_c1_wrw_bn", "dsf_cat_channels_nhwc",
]


def lib():
    """Loads the native library or raises MissingNativeLibrary (never falls back)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise MissingNativeLibrary(
                "libdsf_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(dsf_amd has no CPU fallback)" % LIB_PATH)
        cand = ctypes.CDLL(LIB_PATH)
        rebuild = "rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`"
        try:
            cand.dsf_abi_version.restype = ctypes.c_int
            found = cand.dsf_abi_version()
        except AttributeError:                                      # a library older than the version export
            raise MissingNativeLibrary("%s exports no dsf_abi_version -- %s" % (LIB_PATH, rebuild)) from None
        if found != EXPECTED_ABI:                                   # a stale build: argument lists differ, pointers would shift
            raise MissingNativeLibrary("%s has ABI version %d, this package needs %d -- %s" % (LIB_PATH, found, EXPECTED_ABI, rebuild))
        missing = [s for s in SYMBOLS if not hasattr(cand, s)]
        if missing:                                                 # same version number, fewer exports: still a stale build
            raise MissingNativeLibrary("%s lacks %s -- %s" % (LIB_PATH, ", ".join(missing[:4]), rebuild))
        _lib = cand
        _lib.dsf_status_string.restype = ctypes.c_char_p
        _lib.dsf_conv_x6_image_bytes.restype = ctypes.c_int64
        _lib.dsf_mfma_bf16_probe.restype = ctypes.c_int64
        _lib.dsf_conv_c1_workspace_bytes.restype = ctypes.c_int64
        _lib.dsf_conv_x6_image_granules.restype = ctypes.c_int64
        _lib.dsf_part_volume_workspace_bytes.restype = ctypes.c_int64
        _lib.dsf_conv_x6_wrw_workspace_bytes.restype = ctypes.c_int64
        _lib.dsf_offset2joint_cl_workspace_floats.restype = ctypes.c_int64
    return _lib


def set_deterministic(on=True):
    """Bit-reproducible mode of every kernel of the library (include/dsf_hip.h, "Deterministic mode"); also selected by
    DSF_DETERMINISTIC=1 in the environment.  -> the previous setting."""
    return bool(lib().dsf_set_deterministic(ctypes.c_int(1 if on else 0)))


def deterministic():
    return bool(lib().dsf_get_deterministic())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The current HIP stream of the current device as a launcher argument.  Through torch's raw-stream accessor where it exists:
    torch.cuda.current_stream() builds a Stream object (~9 us; ~280 launches per step call this)"""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_GET_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous GPU tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("dsf_amd ops run on the GPU only (got a %s tensor); there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("dsf_amd ops need contiguous tensors")
    if t.numel() == 0:                                   # empty batch: a valid (never dereferenced) address instead of NULL
        return ctypes.c_void_p(_dummy(t.device).data_ptr())
    return ctypes.c_void_p(t.data_ptr())


def addr(t):
    """Device address of a (possibly strided / empty) GPU tensor's first element, for launchers that take strides."""
    return ctypes.c_void_p(t.data_ptr() if t.numel() else _dummy(t.device).data_ptr())


_DUMMY = {}


def _dummy(device):
    d = _DUMMY.get(device.index)
    if d is None:
        d = _DUMMY[device.index] = torch.zeros(16, device=device)
    return d


def f32(t):
    """Contiguous fp32 view/copy of a GPU tensor (plumbing)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def check(status, what):
    if status != 0:
        raise RuntimeError("%s failed: %s" % (what, lib().dsf_status_string(status).decode()))


def camera(paras, image_size):
    fx, fy, px, py = paras
    return dsf_camera(float(fx), float(fy), float(px), float(py), float(image_size[0]), float(image_size[1]))


F = ctypes.c_float
I = ctypes.c_int
I64 = ctypes.c_int64
