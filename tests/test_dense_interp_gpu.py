"""the dense-query knn_interpolation of kitti_submission.py:89-93 -- every pixel of a 375x1242
disparity map (465,750 queries) against the 8192 input points, k = 3 -- through the product path
(cores.geometry.knn_interpolation -> camli_knn + camli_knn_interp_fwd), checked against the oracle on a random
sample of the queries (every query is independent, so sampled rows are compared exactly / to 1e-5)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dense_query_knn_interpolation_466k_queries(oracle_lib):
    from camliflow_amd.cores import geometry, runtime
    from camliflow_amd.csrc import k_nearest_neighbor
    h, w, n = 375, 1242, 8192
    g = torch.Generator().manual_seed(5)
    f, cx, cy = 721.5, 609.6, 172.9
    disp = torch.rand(h, w, generator=g) * 60 + 3                       # a dense disparity map
    z = 0.54 * f / disp
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    dense = torch.stack([(xs - cx) * z / f, (ys - cy) * z / f, z], dim=0).reshape(1, 3, h * w)     # disp2pc
    pick = torch.randperm(h * w, generator=g)[:n]
    pc1 = dense[:, :, pick].contiguous()                                  # the sparse cloud is a subset of the pixels
    flow = torch.randn(1, 3, n, generator=g) * 0.1
    with torch.no_grad(), runtime.use_backend('hip'):
        runtime.set_strict(True)
        try:
            out = geometry.knn_interpolation(pc1.cuda(), flow.cuda(), dense.cuda(), k=3)
        finally:
            runtime.set_strict(False)
        knn = k_nearest_neighbor(pc1.cuda(), dense.cuda(), 3)
    assert out.shape == (1, 3, h * w) and torch.isfinite(out).all()
    sample = torch.randperm(h * w, generator=g)[:20000]
    sample[:n // 4] = pick[:n // 4]                                       # include coincident query / input points (distance 0)
    q = dense[:, :, sample].contiguous().numpy()
    want_idx = oracle_lib.knn(pc1.numpy().transpose(0, 2, 1), q.transpose(0, 2, 1), 3)
    assert np.array_equal(knn[:, sample].cpu().numpy(), want_idx)
    want = oracle_lib.knn_interp_fwd(pc1.numpy(), flow.numpy(), q, want_idx)
    assert np.allclose(out[:, :, sample].cpu().numpy(), want, rtol=1e-5, atol=1e-6)
