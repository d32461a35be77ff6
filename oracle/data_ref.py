"""CPU restatement of the reference's test-phase depth data path: `Crop_Image_deep_pp` + `normalize_img`
(data/render_loader.py:748-810, 738-745, with `comToBounds` :356-364 and `getCrop` :867-905).  TEST INFRASTRUCTURE ONLY.

Pinning: `tests/golden/reference_data.npz` is produced by importing the reference's `loader` class and calling these very
methods (`tests/golden/make_golden_data.py`).  The one thing the import cannot provide is OpenCV (absent from the image):
`cv2.resize(..., interpolation=cv2.INTER_NEAREST)` is replaced there by `resize_nearest` below, which restates OpenCV's
published `resizeNN` rule (`sx = min(floor(x * (1 / (dw / sw))), sw - 1)`, in double) -- that single function is unpinned.
`com` is taken as float64: the reference hands over float32 `joint3DToImg` output, whose promotion against Python floats
differs between NumPy 1.x (float64 from the first division on) and NumPy >= 2 (float32 throughout); with float64 inputs
both agree, and this file spells the arithmetic out in float64.
"""
import numpy as np


def com_to_bounds(com, size, paras):
    """render_loader.py:356-364"""
    fx, fy, fu, fv = paras
    com = np.asarray(com, dtype=np.float64)
    zstart = com[2] - size[2] / 2.
    zend = com[2] + size[2] / 2.
    xstart = int(np.floor((com[0] * com[2] / fx - size[0] / 2.) / com[2] * fx + 0.5))
    xend = int(np.floor((com[0] * com[2] / fx + size[0] / 2.) / com[2] * fx + 0.5))
    ystart = int(np.floor((com[1] * com[2] / fy - size[1] / 2.) / com[2] * fy + 0.5))
    yend = int(np.floor((com[1] * com[2] / fy + size[1] / 2.) / com[2] * fy + 0.5))
    return xstart, xend, ystart, yend, zstart, zend


def get_crop(depth, xstart, xend, ystart, yend, zstart, zend):
    """render_loader.py:867-905 (2-D branch, thresh_z=True, background 0): out-of-image pixels are 0, values in front of
    the cube are moved to its front face, values behind it become 0."""
    H, W = depth.shape
    out = np.zeros((yend - ystart, xend - xstart), dtype=depth.dtype)
    y0, y1, x0, x1 = max(ystart, 0), min(yend, H), max(xstart, 0), min(xend, W)
    if y1 > y0 and x1 > x0:
        out[y0 - ystart:y1 - ystart, x0 - xstart:x1 - xstart] = depth[y0:y1, x0:x1]
    front = np.logical_and(out < zstart, out != 0)
    back = np.logical_and(out > zend, out != 0)
    out[front] = zstart
    out[back] = 0.
    return out


def resize_nearest(src, dsize):
    """OpenCV `resize(src, (dw, dh), interpolation=INTER_NEAREST)` (imgproc resizeNN): source index =
    min(floor(dst_index * ifx), size - 1) with ifx = 1 / (dw / sw) evaluated in double."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = src.shape
    ifx, ify = 1.0 / (dw / float(sw)), 1.0 / (dh / float(sh))
    xs = np.minimum(np.floor(np.arange(dw) * ifx).astype(np.int64), sw - 1)
    ys = np.minimum(np.floor(np.arange(dh) * ify).astype(np.int64), sh - 1)
    return src[ys][:, xs]


def crop_image_deep_pp(depth, com, size, dsize, paras):
    """render_loader.py:748-810 -> (crop (dsize[1], dsize[0]) float32, 3x3 transform float64)"""
    xstart, xend, ystart, yend, zstart, zend = com_to_bounds(com, size, paras)
    cropped = get_crop(depth, xstart, xend, ystart, yend, zstart, zend)
    wb, hb = xend - xstart, yend - ystart
    if wb > hb:
        sz = (dsize[0], int(hb * dsize[0] / wb))
    else:
        sz = (int(wb * dsize[1] / hb), dsize[1])
    trans = np.eye(3)
    trans[0, 2] = -xstart
    trans[1, 2] = -ystart
    if cropped.shape[0] > cropped.shape[1]:
        scale = np.eye(3) * sz[1] / float(cropped.shape[0])
    else:
        scale = np.eye(3) * sz[0] / float(cropped.shape[1])
    scale[2, 2] = 1
    rz = resize_nearest(cropped, sz)
    ret = np.zeros((dsize[0], dsize[1]), np.float32)       # (the reference builds np.ones(dsize) * 0; dsize is square)
    x0 = int(np.floor(dsize[0] / 2. - rz.shape[1] / 2.))
    y0 = int(np.floor(dsize[1] / 2. - rz.shape[0] / 2.))
    ret[y0:y0 + rz.shape[0], x0:x0 + rz.shape[1]] = rz
    off = np.eye(3)
    off[0, 2] = x0
    off[1, 2] = y0
    return ret, np.dot(off, np.dot(scale, trans))


def normalize_img(premax, img, com, cube):
    """render_loader.py:738-745: the crop's maximum and its zeros go to the far plane, clamp to the cube, scale to [-1, 1].
    float32 image, thresholds in float64 (NumPy 1.x: the in-place subtract / divide run in float32)."""
    img = img.copy()
    far, near = com[2] + (cube[2] / 2.), com[2] - (cube[2] / 2.)
    img[img == premax] = far
    img[img == 0] = far
    img[img >= far] = far
    img[img <= near] = near
    img -= np.float32(com[2])
    img /= np.float32(cube[2] / 2.)
    return img


def crop_and_normalize(depth, com, size, dsize, paras):
    """the `phase == 'test'` branch of loader.__getitem__ (:1909-1916): crop, then normalise with the crop's own maximum"""
    crop, trans = crop_image_deep_pp(depth, com, size, dsize, paras)
    return normalize_img(crop.max(), crop, com, size), trans, crop


# ======================================================================================================================
# training-phase augmentation: `rand_augment` draws are inputs; `augmentCrop` (data/render_loader.py:
# 653-695) = one of moveCoM (:427-456) / rotateHand (:458-497) / scaleHand (:499-527) on the ALREADY CROPPED image, then
# `normalize_img` (:738-745).  The image warps are OpenCV's (`cv2.warpPerspective` inside recropHand :403-424,
# `cv2.getRotationMatrix2D` + `cv2.warpAffine` in rotateHand); OpenCV is absent from the image, so their published
# nearest-neighbour rules (imgproc/imgwarp.cpp) are restated below -- those three functions are UNPINNED, everything around
# them is pinned by running the reference's own methods with these stand-ins (tests/golden/make_golden_data.py).
# Arithmetic convention: float64 throughout, rounded to float32 exactly where the reference stores into float32 arrays
# (`jointImgTo3D` / `joint3DToImg` results) -- NumPy 1.x promotion, the NumPy the reference was written for; under
# NumPy >= 2 the reference's `comToBounds` on such a float32 centre runs in float32 and can move a crop bound by one pixel.
# ======================================================================================================================
AUG_MODES = ('rot', 'com', 'sc', 'none')                   # render_loader.py:1816 (nyu_loader)


def img_to_3d(uvd, paras, flip=1):
    """jointImgTo3D (render_loader.py:288-310): float32 result"""
    fx, fy, fu, fv = paras
    uvd = np.asarray(uvd, dtype=np.float64)
    out = np.stack([(uvd[..., 0] - fu) * uvd[..., 2] / fx, flip * (uvd[..., 1] - fv) * uvd[..., 2] / fy, uvd[..., 2]], -1)
    return out.astype(np.float32)


def to_img(xyz, paras, flip=1):
    """joint3DToImg (render_loader.py:312-333): float32 result"""
    fx, fy, fu, fv = paras
    xyz = np.asarray(xyz, dtype=np.float64)
    out = np.stack([xyz[..., 0] * fx / xyz[..., 2] + fu, flip * xyz[..., 1] * fy / xyz[..., 2] + fv, xyz[..., 2]], -1)
    return out.astype(np.float32)


def com_to_transform(com, size, dsize, paras):
    """comToTransform (render_loader.py:366-401) -> 3x3 float64"""
    xstart, xend, ystart, yend, _, _ = com_to_bounds(com, size, paras)
    wb, hb = xend - xstart, yend - ystart
    if wb > hb:
        s, sz = dsize[0] / float(wb), (dsize[0], hb * dsize[0] / wb)
    else:
        s, sz = dsize[1] / float(hb), (wb * dsize[1] / hb, dsize[1])
    x0 = int(np.floor(dsize[0] / 2. - sz[0] / 2.))
    y0 = int(np.floor(dsize[1] / 2. - sz[1] / 2.))
    return np.array([[s, 0., s * (-xstart) + x0], [0., s, s * (-ystart) + y0], [0., 0., 1.]])


def affine_inverse(M):
    """inverse of a scale-and-shift transform [[s,0,tx],[0,s,ty],[0,0,1]] in closed form, one rounding per entry
    (r = 1/s; -(t r)): what `np.linalg.inv` / `cv::invert` return up to the last ulp, and the form the HIP kernel uses"""
    M = np.asarray(M, dtype=np.float64)
    r = 1.0 / M[0, 0]
    return np.array([[r, 0., -(M[0, 2] * r)], [0., r, -(M[1, 2] * r)], [0., 0., 1.]])


def _cv_round(v):
    """cv::saturate_cast<int>(double) = cvRound: to nearest, ties to even"""
    return np.rint(v).astype(np.int64)


def cv_warp_perspective_nn(src, M, dsize, border=0.0):
    """cv2.warpPerspective(src, M, dsize, flags=INTER_NEAREST, borderMode=BORDER_CONSTANT, borderValue=border):
    M is inverted (dst -> src), the source position of a destination pixel is evaluated in 64 x 16 blocks as
    ((m0 bx + m1 y + m2) + m0 (x - bx)) / w and rounded with cvRound (imgwarp.cpp WarpPerspectiveInvoker)."""
    W, H = int(dsize[0]), int(dsize[1])
    M = np.asarray(M, dtype=np.float64)
    affine = M[0, 1] == 0 and M[1, 0] == 0 and M[0, 0] == M[1, 1] and M[2, 0] == 0 and M[2, 1] == 0 and M[2, 2] == 1
    Mi = affine_inverse(M) if affine else np.linalg.inv(M)
    bh = min(16, H)
    bw = min(1024 // bh, W)
    bh = min(1024 // bw, H)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    bx = np.floor(x / bw) * bw
    X0 = (Mi[0, 0] * bx + Mi[0, 1] * y) + Mi[0, 2]
    Y0 = (Mi[1, 0] * bx + Mi[1, 1] * y) + Mi[1, 2]
    W0 = (Mi[2, 0] * bx + Mi[2, 1] * y) + Mi[2, 2]
    x1 = x - bx
    w = W0 + Mi[2, 0] * x1
    with np.errstate(divide="ignore"):
        w = np.where(w != 0, 1.0 / w, 0.0)
    fx = np.clip((X0 + Mi[0, 0] * x1) * w, -2147483648.0, 2147483647.0)
    fy = np.clip((Y0 + Mi[1, 0] * x1) * w, -2147483648.0, 2147483647.0)
    sx, sy = _cv_round(fx), _cv_round(fy)
    ok = (sx >= 0) & (sx < src.shape[1]) & (sy >= 0) & (sy < src.shape[0])
    out = np.full((H, W), border, dtype=src.dtype)
    out[ok] = src[sy[ok], sx[ok]]
    return out


def cv_get_rotation_matrix_2d(center, angle, scale):
    """cv2.getRotationMatrix2D: center is a Point2f, angle in degrees"""
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))
    a = angle * np.pi / 180.0
    alpha, beta = np.cos(a) * scale, np.sin(a) * scale
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]])


def cv_warp_affine_nn(src, M, dsize, border=0.0):
    """cv2.warpAffine(src, M, dsize, flags=INTER_NEAREST, borderMode=BORDER_CONSTANT): M (2x3) is inverted in double, then
    fixed point with 10 fractional bits: X = (round((m1 y + m2) 1024) + 512 + round(m0 x 1024)) >> 10 (WarpAffineInvoker)."""
    W, H = int(dsize[0]), int(dsize[1])
    m = np.asarray(M, dtype=np.float64).reshape(2, 3).copy()
    D = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[1, 1] * D, m[0, 0] * D
    m[0, 0] = A11; m[0, 1] *= -D; m[1, 0] *= -D; m[1, 1] = A22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    xs, ys = np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64)
    adelta, bdelta = _cv_round(m[0, 0] * xs * 1024), _cv_round(m[1, 0] * xs * 1024)
    X0 = _cv_round((m[0, 1] * ys + m[0, 2]) * 1024) + 512
    Y0 = _cv_round((m[1, 1] * ys + m[1, 2]) * 1024) + 512
    sx = (X0[:, None] + adelta[None, :]) >> 10
    sy = (Y0[:, None] + bdelta[None, :]) >> 10
    ok = (sx >= 0) & (sx < src.shape[1]) & (sy >= 0) & (sy < src.shape[0])
    out = np.full((H, W), border, dtype=src.dtype)
    out[ok] = src[sy[ok], sx[ok]]
    return out


def recrop_hand(crop, M, Mnew, target, paras, background, nv_val, com, size):
    """recropHand (render_loader.py:403-424), thresh_z=True"""
    warped = cv_warp_perspective_nn(crop, np.dot(M, Mnew), target, border=float(background))
    warped[warped < nv_val] = background
    _, _, _, _, zstart, zend = com_to_bounds(com, size, paras)
    front = np.logical_and(warped < zstart, warped != 0)
    back = np.logical_and(warped > zend, warped != 0)
    warped[front] = zstart
    warped[back] = 0.
    return warped


def augment_crop(img, joints, com, cube, M, mode, off, rot, sc, paras, flip=1):
    """augmentCrop (render_loader.py:653-695).  img (S,S) float32 raw crop (mm, 0 = background), joints (J,3) float32 relative
    to the centre, com (3,) image coordinates (u, v, z), cube (3,), M (3,3), mode in AUG_MODES, off (3,) mm, rot degrees, sc.
    -> (normalised image float32, joints float32, cube (3,) float64, com (3,), M (3,3))"""
    img = np.asarray(img, dtype=np.float32)
    joints = np.asarray(joints, dtype=np.float32)
    com = np.asarray(com)
    cube = [float(c) for c in cube]
    S = img.shape[0]
    premax = img.max()
    new_img, new_joints = img, joints
    if img.max() == 0 or mode == 'none':
        pass
    elif mode == 'com' and not np.allclose(off, 0.):
        new_com = to_img(img_to_3d(com, paras, flip).astype(np.float64) + np.asarray(off, dtype=np.float64), paras, flip)
        if not (np.allclose(com[2], 0.) or np.allclose(new_com[2], 0.)):
            Mnew = com_to_transform(new_com.astype(np.float64), cube, img.shape, paras)
            new_img = recrop_hand(img.copy(), Mnew, affine_inverse(M), img.shape, paras, 0, img[img > 0].min() - 1, new_com.astype(np.float64), cube)
        else:
            Mnew = M
        new_joints = (joints + img_to_3d(com, paras, flip) - img_to_3d(new_com, paras, flip)).astype(np.float32)
        com, M = new_com, Mnew
    elif mode == 'rot' and not np.allclose(rot, 0.):
        r = np.mod(rot, 360)
        R = cv_get_rotation_matrix_2d((S // 2, S // 2), -r, 1)
        new_img = cv_warp_affine_nn(img, R, (S, S), border=0.0)
        if (img > 0).sum() > 0:
            new_img[new_img < img[img > 0].min() - 1] = 0
        com3d = img_to_3d(com, paras, flip)
        j2d = to_img(joints + com3d, paras, flip).astype(np.float64)         # float32 values, rotated in double
        a = r * np.pi / 180.
        c2 = np.asarray(com[0:2], dtype=np.float64)
        # rotatePoint2D (:130-147) works in place on float32 rows: every statement rounds to float32
        pp = (j2d[:, 0:2] - c2).astype(np.float32).astype(np.float64)
        rotd = np.stack([pp[:, 0] * np.cos(a) - pp[:, 1] * np.sin(a), pp[:, 0] * np.sin(a) + pp[:, 1] * np.cos(a)], -1)
        rotd = (rotd.astype(np.float32).astype(np.float64) + c2).astype(np.float32)
        d2 = np.concatenate([rotd, j2d[:, 2:3].astype(np.float32)], -1)
        new_joints = (img_to_3d(d2, paras, flip) - com3d).astype(np.float32)
    elif mode == 'sc' and not np.allclose(sc, 1.):
        new_cube = [s * sc for s in cube]
        if not np.allclose(com[2], 0.):
            Mnew = com_to_transform(np.asarray(com, dtype=np.float64), new_cube, img.shape, paras)
            new_img = recrop_hand(img.copy(), Mnew, affine_inverse(M), img.shape, paras, 0, img[img > 0].min() - 1,
                                  np.asarray(com, dtype=np.float64), cube)
            M = Mnew
        cube = new_cube
    out = normalize_img(premax, new_img, np.asarray(com, dtype=np.float64), cube)
    return out, new_joints, np.asarray(cube, dtype=np.float64), np.asarray(com), np.asarray(M, dtype=np.float64)
