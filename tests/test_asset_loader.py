"""The real-asset path of the MANO loader (dsf_amd/assets.py::load_mano_dict + _ChShim), which every other test bypasses with
'synthetic'.  The licensed MANO_RIGHT.pkl is a Python-2 pickle whose arrays are `chumpy.ch.Ch` objects
(/root/reference/render_model/mano_layer.py:98-160 reads it with pickle.load(..., encoding='latin1') and np.array(model[key])).
chumpy is not installed here, so the test writes a file of exactly that shape -- the synthetic hand with its arrays wrapped in a
fake `chumpy.ch.Ch` (protocol 2, the attribute set chumpy's objects carry) -- removes the fake module again and loads the file
through the product's loader."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

CH_KEYS = ("v_template", "shapedirs", "posedirs", "weights", "J")


def _write_chumpy_style_pickle(path, model):
    chumpy = types.ModuleType("chumpy")
    ch = types.ModuleType("chumpy.ch")

    class Ch(object):                                  # pickled by reference: module chumpy.ch, name Ch
        def __init__(self, x):
            self.x = np.asarray(x)
            self._dirty_vars = set()
            self._itr = None
            self._parents = {}
            self._cache = {"r": None, "drs": {}}
            self._depends_on_deps = None
    Ch.__module__ = "chumpy.ch"
    Ch.__qualname__ = "Ch"
    ch.Ch = Ch
    chumpy.ch = ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = chumpy, ch
    try:
        wrapped = dict(model)
        wrapped["J"] = np.asarray(model["J_regressor"].dot(model["v_template"]))
        for k in CH_KEYS:
            wrapped[k] = Ch(wrapped[k])
        wrapped["bs_style"], wrapped["bs_type"] = "lbs", "lrotmin"
        with open(path, "wb") as fh:
            pickle.dump(wrapped, fh, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]


@pytest.fixture()
def mano_dir(tmp_path, mano_dict):
    d = tmp_path / "mano"
    d.mkdir()
    _write_chumpy_style_pickle(str(d / "MANO_RIGHT.pkl"), mano_dict)
    return str(d)


def test_pickle_really_references_chumpy(mano_dir):
    raw = open(os.path.join(mano_dir, "MANO_RIGHT.pkl"), "rb").read()
    assert b"chumpy.ch" in raw and b"Ch" in raw
    assert "chumpy" not in sys.modules
    with pytest.raises(ModuleNotFoundError):            # the plain unpickler the reference uses needs chumpy installed
        pickle.load(open(os.path.join(mano_dir, "MANO_RIGHT.pkl"), "rb"), encoding="latin1")


def test_load_mano_dict_reads_a_chumpy_pickle(mano_dir, mano_dict):
    from dsf_amd.assets import load_mano_dict
    got = load_mano_dict(os.path.join(mano_dir, "MANO_RIGHT.pkl"))
    for k in ("f", "v_template", "shapedirs", "posedirs", "weights", "hands_components", "hands_mean", "kintree_table"):
        a, b = np.array(got[k], dtype=np.float64), np.array(mano_dict[k], dtype=np.float64)
        assert a.shape == b.shape and np.array_equal(a, b), k
    assert np.array_equal(got["J_regressor"].toarray(), mano_dict["J_regressor"].toarray())
    assert np.array(got["shapedirs"]).shape == (778, 3, 10)                      # the Ch objects convert like arrays (:116)


def test_mano_layer_from_a_chumpy_pickle_equals_the_synthetic_one(mano_dir):
    from dsf_amd.render_model.mano_layer import MANO_SMPL
    a = MANO_SMPL(os.path.join(mano_dir, "MANO_RIGHT.pkl"), "nyu")
    b = MANO_SMPL("synthetic", "nyu")
    sa, sb = dict(a.named_buffers()), dict(b.named_buffers())
    assert sorted(sa) == sorted(sb) and len(sa) >= 8
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(a.faces, b.faces) and np.array_equal(a.parents, b.parents)


def test_missing_asset_is_an_error_not_a_silent_synthetic_hand(tmp_path):
    from dsf_amd.assets import load_mano_dict
    with pytest.raises(FileNotFoundError):
        load_mano_dict(str(tmp_path / "nowhere" / "MANO_RIGHT.pkl"))


@pytest.mark.gpu
def test_render_constructed_from_a_chumpy_pickle_renders_like_the_synthetic_one(mano_dir):
    from dsf_amd.render_model.mano_layer import Render
    cam = (588.03, 587.07, 320.0, 240.0)
    # Render(mano_path, ...) takes the directory, as the reference does (mano_layer.py:972)
    r_file = Render(mano_dir, "nyu", cam, (640, 480)).cuda()
    r_syn = Render("synthetic", "nyu", cam, (640, 480)).cuda()
    g = torch.Generator().manual_seed(5)
    P = torch.zeros(3, 62)
    P[:, :3] = torch.rand(3, 3, generator=g) * 2 - 1
    P[:, 3:58] = torch.randn(3, 55, generator=g) * 0.4
    P[:, 58] = 1.0
    center = torch.tensor([[10.0, -5.0, 700.0], [-20.0, 15.0, 900.0], [0.0, 0.0, 600.0]])
    cube = torch.full((3, 3), 250.0)
    outs_f = r_file.render(P.cuda(), center.cuda(), cube.cuda())
    outs_s = r_syn.render(P.cuda(), center.cuda(), cube.cuda())
    for a, b in zip(outs_f, outs_s):
        assert torch.equal(a, b)
    assert (outs_f[0] < 0.99).float().mean().item() > 0.02
