"""oracle/glue.py (numpy restatement of the GRU2D gate arithmetic, SKFusion, the fusion-aware score product and the
l2-norm sequence loss, forward + hand-written adjoints) pinned on what the REFERENCE's own modules and autograd produced
(tests/golden/glue_*.npz, tests/golden/make_glue_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import glue

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


def close(a, b, tol=2e-6):
    return np.allclose(a, b, rtol=1e-5, atol=tol)


def test_gru2d_forward_and_adjoint_chain():
    """Both half-steps of the reference's GRU2D.forward (raft_core.py:124-138): the oracle's gates / blend reproduce
    every recorded intermediate, and its adjoints, chained with the recorded convolution data-gradients, reproduce
    the gradients autograd left on the convolution outputs and on h."""
    g = load('glue_gru2d')
    c = int(g['hidden'])
    zero = np.zeros_like(g['convz1_out'])
    zero2 = np.concatenate([zero, zero], 1)
    h = g['h0']
    fw = []
    for half in ('1', '2'):
        pre_zr = np.concatenate([g['convz%s_out' % half], g['convr%s_out' % half]], 1)
        z, rh, r = glue.gru_gates_fwd(pre_zr, zero2, h)
        assert close(rh, g['convq%s_in' % half][:, :c])
        h_new, q = glue.gru_blend_fwd(g['convq%s_out' % half], zero, z, h, nan_to_num=(half == '2'))
        fw.append((z, r, h, q))
        if half == '1':
            assert close(h_new, g['convz2_in'][:, :c])
        else:
            assert close(h_new, g['out'])
        h = h_new
    grad = g['gout']
    for half, (z, r, h_in, q) in (('2', fw[1]), ('1', fw[0])):
        gpre_q, gz, gh_blend = glue.gru_blend_bwd(grad, z, h_in, q)
        assert close(gpre_q, g['convq%s_out_grad' % half])
        grh = g['convq%s_in_grad' % half][:, :c]
        gpre_zr, gh_gates = glue.gru_gates_bwd(gz, grh, z, r, h_in)
        assert close(gpre_zr[:, :c], g['convz%s_out_grad' % half])
        assert close(gpre_zr[:, c:], g['convr%s_out_grad' % half])
        grad = gh_blend + gh_gates + g['convz%s_in_grad' % half][:, :c]      # hx feeds convz AND convr: one tensor, one .grad
    assert close(grad, g['h0_grad'], 1e-5)


@pytest.mark.parametrize('fmt', ['nchw', 'ncm'])
def test_skfusion_forward_and_adjoint(fmt):
    g = load('glue_skfusion_' + fmt)
    a, b = g['a'], g['b']
    s = glue.sk_pool_fwd(a, b)
    assert close(s, g['s'])
    w = glue.sk_gate_fwd(s, g['wmid'], g['wout'])
    assert close(glue.sk_mix_fwd(a, b, w), g['out'])
    _, _, gw = glue.sk_fuse_bwd(g['gout'], a, b, w, np.zeros_like(s))
    gs, gwmid, gwout = glue.sk_gate_bwd(gw, s, g['wmid'], g['wout'])
    assert close(gs, g['s_grad'])
    assert close(gwmid, g['wmid_grad'], 1e-5) and close(gwout, g['wout_grad'], 1e-5)
    ga, gb, _ = glue.sk_fuse_bwd(g['gout'], a, b, w, gs)
    assert close(ga, g['a_grad']) and close(gb, g['b_grad'])


def test_gather_scale():
    g = load('glue_gather_scale')
    out, gathered = glue.gather_scale_fwd(g['data'], g['score'], g['idx'])
    assert np.array_equal(out, g['out'])
    assert np.array_equal(glue.gather_scale_bwd_score(g['gout'], gathered), g['score_grad'])


@pytest.mark.parametrize('name', ['2d_mask', '2d_nomask', '3d_mask', '3d_nomask'])
def test_sequence_loss_l2(name):
    g = load('glue_seqloss_' + name)
    preds = [g['pred%d' % i] for i in range(3)]
    c, gamma = int(g['n_channels']), float(g['gamma'])
    assert abs(glue.sequence_loss_l2_fwd(preds, g['target'], c, gamma) - g['loss']) <= 1e-6 * abs(g['loss'])
    for got, i in zip(glue.sequence_loss_l2_bwd(preds, g['target'], c, gamma), range(3)):
        assert np.allclose(got, g['grad%d' % i], rtol=1e-5, atol=1e-9)
