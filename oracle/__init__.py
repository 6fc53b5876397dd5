"""CPU oracle for the depth -> world point-cloud fusion path.  TEST INFRASTRUCTURE ONLY.

This package restates, in NumPy / plain Python, the arithmetic of the reference's
hot path (rainfall1998/3D_reconstruction_system: transfer/pixel_to_camera.py,
transfer/camera_to_world.py, other_tools/transfer_T_icp.py).  It exists so that
tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg can CHECK and
TIME-BESIDE the HIP path.  Nothing in `3d_reconstruction_system_amd/` (the product)
imports it; the product fails loudly when its HIP library is missing.

Parity status
-------------
* fusion_ref (unproject, quaternion->R^-1, SE(3) apply, frame fusion, apply-T, txt/PLY
  formats): PINNED.  Checked against fixtures in tests/golden/ that were produced by
  importing and running the unmodified reference (tests/golden/make_golden.py).
* icp_ref (nearest-neighbour + cross-covariance + Umeyama): PARITY UNPINNED.  The
  reference contains no ICP estimation code (its transfer_T_icp.py only applies a
  pre-computed T_data.txt); the restated algorithm is the published Umeyama (1991)
  closed form with brute-force squared-L2 NN, anchored only on the consumer side
  (get_T file format + apply), and on synthetic known-answer recoveries.
"""
