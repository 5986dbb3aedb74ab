"""CPU oracle (TEST INFRASTRUCTURE -- see camli_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
``camliflow_amd`` never does.  Parity pin: tests/golden/*.npz (generated from the reference's own
Python path by tests/golden/make_golden.py) are checked against these functions in
tests/test_oracle_golden.py.
"""
from .binding import (build, knn, fps, corr2d_fwd, corr2d_bwd, allpairs_lookup_fwd, allpairs_lookup_bwd,
                      gather_cf, scatter_add_cf, gather_cl, scatter_add_cl, knn_interp_fwd, pointconv_dw_fwd, pointconv_dw_bwd, knn_interp_bwd, knn_interp_bwd_xyz, pwc3d_pair_fwd, ksum_fwd, gather_wsum_fwd, persp2paral, pad_normalize, project_pc2image,
                      corr3d_gather_fwd, pointconv_mix_fwd, convex_upsample_fwd, weightnet_fwd, weightnet_bwd, bilinear_sample_fwd, ids_flow_fwd,
                      LIB_PATH)
