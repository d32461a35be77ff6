"""Golden vectors for the depth data path: runs the REFERENCE's own `loader.Crop_Image_deep_pp` and
`loader.normalize_img` (data/render_loader.py) on synthetic 480x640 depth frames.  Run in the build container only
(`/root/reference` is not on the GPU box):  python tests/golden/make_golden_data.py  ->  tests/golden/reference_data.npz

OpenCV is absent from the image; `cv2.resize(.., interpolation=cv2.INTER_NEAREST)` is provided by `oracle.data_ref.
resize_nearest` (OpenCV's published resizeNN index rule) -- the only step of these vectors that is not the reference's code.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # stubs for the third-party modules the container lacks

PARAS = (588.03, 587.07, 320.0, 240.0)


def frames(rng, n):
    """depth frames in mm: far background with holes (0), a hand-sized blob around the centre of mass, some pixels in
    front of and behind the crop cube"""
    out, coms, cubes = [], [], []
    for i in range(n):
        z = rng.uniform(450, 1100)
        u, v = rng.uniform(40, 600), rng.uniform(40, 440)
        if i % 4 == 3:                                    # crop box leaves the image
            u, v = rng.choice([8.0, 632.0]), rng.choice([6.0, 474.0])
        d = np.full((480, 640), 0, np.float32)
        bg = rng.uniform(z + 200, z + 900)
        d[:] = bg + rng.normal(0, 3, d.shape)
        d[rng.uniform(size=d.shape) < 0.03] = 0
        yy, xx = np.mgrid[0:480, 0:640]
        r = 70 * 550.0 / z
        blob = (xx - u) ** 2 + (yy - v) ** 2 < r * r
        d[blob] = (z + 40 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + rng.normal(0, 2, d.shape))[blob]
        d[int(min(max(v, 0), 479)), :] = z - 400          # a line in front of the cube
        cube = float(rng.choice([250, 300, 200]))
        out.append(d.astype(np.float32)); coms.append([u, v, z]); cubes.append([cube, cube, cube])
    return np.stack(out), np.asarray(coms, np.float64), np.asarray(cubes, np.float64)


def main():
    mg._install_stubs()
    sys.path.insert(0, mg.REPO)
    from oracle import data_ref
    import cv2
    cv2.INTER_NEAREST = 0
    cv2.resize = lambda src, dsize, interpolation=0: data_ref.resize_nearest(src, dsize)
    sys.path.insert(0, mg.REF)
    from data import render_loader as rl
    L = rl.loader("/x", "test", 128, "refine", "nyu")
    L.paras = PARAS
    rng = np.random.RandomState(11)
    depth, com, cube = frames(rng, 12)
    crops, trans, norm = [], [], []
    for d, c, s in zip(depth, com, cube):
        crop, M = L.Crop_Image_deep_pp(d.copy(), c, s, (128, 128), PARAS)
        crops.append(crop.copy())
        trans.append(M)
        norm.append(L.normalize_img(crop.max(), crop.copy(), c, s))
    # keep the fixture small: frames are regenerated from the seed by the tests (frames() is imported from this file)
    np.savez_compressed(os.path.join(HERE, "reference_data.npz"), com=com, cube=cube, crop=np.stack(crops).astype(np.float32),
                        trans=np.stack(trans), norm=np.stack(norm).astype(np.float32), seed=np.int64(11))
    print("wrote reference_data.npz", np.stack(crops).shape, os.path.getsize(os.path.join(HERE, "reference_data.npz")))


if __name__ == "__main__":
    main()
