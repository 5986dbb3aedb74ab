"""Golden vectors for oracle/glue.py from the REFERENCE's own modules (GRU2D, SKFusion, batch_indexing,
calc_sequence_loss_2d / _3d), run unchanged with autograd on seeded inputs; intermediates are captured with module
hooks (conv outputs / inputs with retain_grad), so that the hand-written adjoints of oracle/glue.py can be chained
through one GRU2D.forward / SKFusion.forward and compared with what autograd produced.

Run in the build container only:  python tests/golden/make_glue_golden.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import refmodels  # noqa: E402

refmodels.install(native_semantics=True)
from models.clfm import SKFusion  # noqa: E402
from models.losses import calc_sequence_loss_2d, calc_sequence_loss_3d  # noqa: E402
from models.raft_core import GRU2D  # noqa: E402
from models.utils import batch_indexing  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print('%-28s %7.1f KB' % (name, os.path.getsize(path) / 1024))


def keep_io(mod, store, key):
    """record the module's input and output tensors of its one call (with their gradients after backward)"""
    def pre(_m, args):
        args[0].retain_grad()
        store[key + '_in'] = args[0]

    def post(_m, _args, out):
        out.retain_grad()
        store[key + '_out'] = out
    mod.register_forward_pre_hook(pre)
    mod.register_forward_hook(post)


def golden_gru():
    g = torch.Generator().manual_seed(11)
    c, cin, b, hh, ww = 8, 12, 2, 6, 7
    gru = GRU2D(hidden_dim=c, input_dim=cin)
    for p in gru.parameters():
        p.data = torch.randn(p.shape, generator=g) * 0.3
    store = {}
    for n in ('convz1', 'convr1', 'convq1', 'convz2', 'convr2', 'convq2'):
        keep_io(getattr(gru, n), store, n)
    h0 = torch.randn(b, c, hh, ww, generator=g, requires_grad=True)
    x = torch.randn(b, cin, hh, ww, generator=g, requires_grad=True)
    gout = torch.randn(b, c, hh, ww, generator=g)
    out = gru(h0, x)
    out.backward(gout)
    arrays = {'h0': h0, 'x': x, 'gout': gout, 'out': out, 'h0_grad': h0.grad, 'hidden': c}
    for k, t in store.items():
        arrays[k] = t
        arrays[k + '_grad'] = t.grad
    save('glue_gru2d', **arrays)


def golden_skfusion():
    for fmt, shape in (('nchw', (2, 6, 5, 7)), ('ncm', (3, 6, 33))):
        g = torch.Generator().manual_seed(23 + len(shape))
        cin = shape[1]
        mod = SKFusion(cin, cin, 8, fmt, None, reduction=2)
        for p in mod.parameters():
            p.data = torch.randn(p.shape, generator=g) * 0.4
        store = {}
        keep_io(mod.align1, store, 'align1')
        keep_io(mod.align2, store, 'align2')
        keep_io(mod.fc_mid, store, 'fc_mid')
        x = torch.randn(*shape, generator=g, requires_grad=True)
        y = torch.randn(*shape, generator=g, requires_grad=True)
        out = mod(x, y)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        save('glue_skfusion_' + fmt, a=store['align1_out'], b=store['align2_out'], a_grad=store['align1_out'].grad,
             b_grad=store['align2_out'].grad, s=store['fc_mid_in'], s_grad=store['fc_mid_in'].grad,
             wmid=mod.fc_mid[0].weight, wout=mod.fc_out[0].weight, wmid_grad=mod.fc_mid[0].weight.grad,
             wout_grad=mod.fc_out[0].weight.grad, out=out, gout=gout)


def golden_gather_scale():
    g = torch.Generator().manual_seed(5)
    b, c, m, p = 2, 5, 40, 63
    data = torch.randn(b, c, m, generator=g)
    score = torch.rand(b, c, p, 1, generator=g, requires_grad=True)
    idx = torch.randint(0, m, (b, p, 1), generator=g)
    final = (score * batch_indexing(data, idx)).sum(dim=-1)          # clfm.py:62-76 with k = 1
    gout = torch.randn(final.shape, generator=g)
    final.backward(gout)
    save('glue_gather_scale', data=data, score=score[..., 0], idx=idx[..., 0], out=final, gout=gout,
         score_grad=score.grad[..., 0])


def golden_sequence_loss():
    cfgs = SimpleNamespace(gamma=0.8, order='l2-norm')
    for name, fn, c, sp in (('2d', calc_sequence_loss_2d, 2, (9, 11)), ('3d', calc_sequence_loss_3d, 3, (37,))):
        for masked in (True, False):
            g = torch.Generator().manual_seed(31 + c + int(masked))
            target = torch.randn(2, c + int(masked), *sp, generator=g)
            if masked:
                target[:, c] = (torch.rand(2, *sp, generator=g) > 0.3).float()
            preds = [torch.randn(2, c, *sp, generator=g).requires_grad_(True) for _ in range(3)]
            with torch.no_grad():
                preds[1][0, :, ..., :2] = target[0, :c, ..., :2]        # exact hits
            loss = fn(preds, target, cfgs)
            loss.backward()
            save('glue_seqloss_%s_%s' % (name, 'mask' if masked else 'nomask'), target=target, loss=loss,
                 n_channels=c, gamma=cfgs.gamma, **{'pred%d' % i: q for i, q in enumerate(preds)},
                 **{'grad%d' % i: q.grad for i, q in enumerate(preds)})


if __name__ == '__main__':
    golden_gru()
    golden_skfusion()
    golden_gather_scale()
    golden_sequence_loss()
