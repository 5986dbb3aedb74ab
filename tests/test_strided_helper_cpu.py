"""fused._batch_strided: which gradient tensors the *_strided entry points may read in place (host logic, no GPU needed)."""
import torch

from camliflow_amd.csrc import fused


def test_contiguous_and_channel_slices_are_accepted_with_their_batch_stride():
    wide = torch.zeros(3, 12, 4, 8)
    assert fused._batch_strided(wide) == 12 * 32
    assert fused._batch_strided(wide[:, 4:8]) == 12 * 32              # the adjoint of a cat hands these over
    assert fused._batch_strided(wide[:, :5]) == 12 * 32
    assert fused._batch_strided(torch.zeros(1, 6, 16)[:, 2:4]) == 2 * 16      # batch 1: any stride would do, the dense one is reported
    assert fused._batch_strided(torch.zeros(2, 8, 16, 1)[:, :4]) == 8 * 16     # size-1 axes carry no layout


def test_everything_else_is_refused():
    wide = torch.zeros(3, 12, 4, 8)
    assert fused._batch_strided(wide[:, :, 1:3]) is None                # rows cropped: planes are not dense
    assert fused._batch_strided(wide[..., ::2]) is None
    assert fused._batch_strided(wide.permute(0, 2, 3, 1)) is None       # channels-last view
    assert fused._batch_strided(wide.double()) is None
    assert fused._batch_strided(torch.zeros(3)) is None
    odd = torch.zeros(2, 3, 5, 7)
    assert fused._batch_strided(odd) is None                            # batch stride 105 is not a multiple of 4 floats
    assert fused._batch_strided(wide[:, 1:5].reshape(3, 4, 32)) == 12 * 32          # merging the trailing axes keeps it a view
    shifted = torch.zeros(2 * 12 * 32 + 1)[1:].view(2, 12, 32)
    assert fused._batch_strided(shifted) is None                        # not 16-byte aligned
