"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot path (naver/dust3r @ /root/reference):
  croco_ref/models/*   restated naver/croco modules (submodule absent -> PARITY UNPINNED)
  roma_ref.py          restated `roma` subset           (dependency absent -> PARITY UNPINNED)
  dust3r_ref.py        restated dust3r glue: model.py / heads / postprocess / inference
  aligner_ref.py       restated cloud_opt PointCloudOptimizer forward + Adam loop
  ref_import.py        (build container only) imports the UNMODIFIED reference files from
                       /root/reference on top of the restated croco/roma shims, to pin the
                       restatements above and to generate tests/golden/* (make_golden.py)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package. The product (dust3r_amd/) never imports it and fails loudly without its HIP library.
"""
