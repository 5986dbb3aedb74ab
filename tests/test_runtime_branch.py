"""runtime.Branch off the GPU / with the lanes off: a transparent no-op (one stream, same results), so the CPU mirrors and the
one-lane configurations are exactly what they were.  The enabled path is covered on the GPU by
tests/test_model_gpu.py::test_two_lane_stream_overlap_matches_single_stream (its passes after the priming one run the point
lane AND the auxiliary streams) and by the full-size parity tests."""
import torch

from camliflow_amd.cores import runtime


def test_branch_is_a_no_op_without_cuda_lanes():
    x = torch.randn(3, 4)
    branch = runtime.Branch(x, None, [x, None], slot=2)
    assert not branch.enabled
    with branch as b:
        assert b is branch
        y = x * 2
    branch.join(y, None)
    assert torch.equal(y, x * 2)


def test_motion_encoder_begin_handle_matches_plain_forward_on_cpu():
    """begin() returns None off the fused path and forward() then computes the flow branch itself; a handle produced by an
    (inactive) Branch gives the same values."""
    from camliflow_amd.cores.raft2d import MotionEncoder2D
    torch.manual_seed(0)
    enc = MotionEncoder2D(4, 4)
    flow, corr = torch.randn(1, 2, 8, 10), torch.randn(1, 4 * 81, 8, 10)
    assert enc.begin(flow) is None
    want = enc(flow, corr)
    f = enc.relu(enc.conv_f2(enc.relu(enc.conv_f1(flow))))
    got = enc(flow, corr, flow_branch=(runtime.Branch(flow), f))
    assert torch.allclose(got, want, atol=1e-6)


def test_set_overlap_false_switches_the_auxiliary_streams_off():
    """_LANES_LIVE is set by the Lanes of a pass; a model that builds no Lanes (the image-only RAFT, CamLiPWC) must not
    inherit it once the overlap is switched off."""
    runtime._LANES_LIVE = True
    try:
        runtime.set_overlap(False)
        assert runtime._LANES_LIVE is False
    finally:
        runtime._LANES_LIVE = False


def test_auxiliary_streams_do_not_outlive_the_pass_that_enabled_them():
    """With the overlap left on, the flag set by one pass's Lanes must not reach a later model that builds none: it is tied
    to the lifetime of the Lanes object (runtime.lanes_live)."""
    import weakref

    class Pass:
        pass

    saved = runtime._LANES_LIVE, runtime._LANES_OWNER
    try:
        owner = Pass()
        runtime._LANES_LIVE, runtime._LANES_OWNER = True, weakref.ref(owner)
        assert runtime.lanes_live()
        del owner
        assert not runtime.lanes_live()
        runtime._LANES_LIVE, runtime._LANES_OWNER = False, None
        assert not runtime.lanes_live()
        lanes = runtime.Lanes(torch.device('cpu'))      # a real Lanes registers itself (disabled off the GPU)
        assert runtime._LANES_OWNER() is lanes and not runtime.lanes_live()
    finally:
        runtime._LANES_LIVE, runtime._LANES_OWNER = saved


def test_upsampler_begin_finish_equal_forward_on_cpu():
    """begin() yields no handle off the fused path; finish(None, h, flow) is forward(h, flow)."""
    from camliflow_amd.cores.raft2d import ConvexUpsampler2D
    torch.manual_seed(1)
    up = ConvexUpsampler2D(16)
    h, flow = torch.randn(1, 16, 6, 7), torch.randn(1, 2, 6, 7)
    assert up.begin(h) is None
    assert torch.equal(up.finish(None, h, flow), up(h, flow))
