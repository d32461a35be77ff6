"""CPU tier: static checks on the gfx950 code objects of the shipped library (dsf_amd/csrc/isa_lint.py).

Round 4 built the library with -fno-slp-vectorize after the auto-vectorised MANO backward returned wrong bits beside convolution
workgroups, and said so in build.sh -- while the loop vectoriser still put packed-FP32 instructions into four other kernels
(the crop rasteriser among them).  This test disassembles what is actually shipped."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dsf_amd", "csrc"))
import isa_lint  # noqa: E402

HIPCC = "/opt/rocm/bin/hipcc"


def _lib_path():
    from dsf_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.LIB_PATH


def test_every_code_object_is_free_of_scratch_and_packed_fp32():
    bad, summary = isa_lint.lint(_lib_path())
    assert summary["code_objects"] == 16, summary                  # one per .hip source of build.sh
    assert summary["kernels"] >= 100 and summary["instructions"] > 100000, summary
    assert not bad, "\n".join(bad)


def test_build_flags_still_disable_both_vectorisers():
    flags = open(os.path.join(REPO, "dsf_amd", "lib", ".flags")).read()
    assert "-fno-slp-vectorize" in flags and "-fno-vectorize" in flags and "-ffp-contract=off" in flags


@pytest.mark.skipif(not os.path.isfile(HIPCC), reason="needs hipcc")
def test_lint_sees_what_it_is_meant_to_see(tmp_path):
    """The checker itself: a kernel the SLP vectoriser packs (and that overwrites a packed source right behind the op) and a
    kernel with a stack array must both be reported."""
    src = tmp_path / "probe.hip"
    src.write_text(r'''
#include <hip/hip_runtime.h>
__global__ void packed(const float* __restrict__ a, float* __restrict__ o, int n) {
    float x0 = a[threadIdx.x], x1 = a[threadIdx.x + 64], y0 = 0.f, y1 = 0.f;
    for (int i = 0; i < n; ++i) { y0 = fmaf(x0, 1.5f, y0); y1 = fmaf(x1, 1.5f, y1); x0 = y1 * 0.5f; x1 = y0 * 0.25f; }
    o[threadIdx.x * 2] = y0; o[threadIdx.x * 2 + 1] = y1;
}
__global__ void spills(const int* __restrict__ idx, float* __restrict__ o) {
    float t[64];
    for (int i = 0; i < 64; ++i) t[i] = i * 0.5f;
    for (int i = 0; i < 64; ++i) t[idx[i] & 63] += 1.f;
    o[threadIdx.x] = t[idx[threadIdx.x] & 63];
}
''')
    so = tmp_path / "libprobe.so"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", str(src), "-o", str(so)], stderr=subprocess.DEVNULL)
    bad, summary = isa_lint.lint(str(so))
    assert summary["code_objects"] == 1
    text = "\n".join(bad)
    assert "packed" in text and "packed-FP32" in text, text
    assert "spills" in text and "scratch" in text, text


def test_write_after_read_detector_on_text():
    ins = ["v_pk_fma_f32 v[8:9], v[4:5], v[20:21], v[8:9] op_sel_hi:[0,1,1]", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v20, v25",
           "v_pk_mul_f32 v[2:3], v[2:3], v[6:7]", "v_add_f32_e32 v10, v2, v9"]
    pairs = isa_lint.war_pairs(ins)
    assert len(pairs) == 1 and pairs[0][1] == "v_mov_b32_e32 v20, v25"
