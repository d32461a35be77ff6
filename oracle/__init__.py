"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference algorithm for the hot path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / the reported CPU baseline.  Nothing under
dsf_amd/ imports it; the product path raises when the HIP library is missing.

  hand_ref.py   MANO layer, spheres, collision, segmentation   (torch-CPU fp32)
  image_ref.py  crop chain, index maps, loader utils, GFM, losses (numpy/torch)
  p3d_ref.c     pytorch3d==0.4.0 rasteriser + point-face distance (plain C)
  p3d.py        ctypes binding of p3d_ref.c
  eval_ref.py   evaluation metric (Trainer.xyz2error) and joint selection (numpy)
  data_ref.py   test-phase depth crop + normalisation of the data loader (numpy)
  nets.py       torch.nn twins of the product networks (layer registry swapped in a context manager)
  step_ref.py   one whole BASELINE-config-2 step from the pieces above (end-to-end checker, cpu_baseline)
  Makefile      builds oracle/_build/liboracle_p3d.so with gcc

Pinning: hand_ref / image_ref / eval_ref / nets are checked against golden vectors produced by
importing /root/reference (tests/golden/make_golden.py, make_golden_eval.py, make_golden_data.py; data_ref's
nearest-neighbour resize restates OpenCV's rule, OpenCV being absent: that function alone is unpinned).  p3d_ref.c restates a
third-party dependency (pytorch3d==0.4.0, pinned in /root/reference/README.md:41,
not vendored, not installable here): PARITY UNPINNED against the real wheel;
anchored on known-answer tests (tests/test_oracle_p3d.py).

oracle/_ref (a build of the reference's own sources) does not exist: the
reference is pure Python and its native layer is the absent pytorch3d wheel.
"""
