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
