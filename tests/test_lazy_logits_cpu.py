"""CPU (emulated kernels): the lazily materialised ``pred_segmap`` (dcs_amd.losses.LazyLogits, network/utils.py:8 +
weathernet.py:93 in the reference).  The criteria evaluate the handle fused; any other use sees an ordinary tensor."""
import pytest
import torch

import emu_ops
from oracle import swiftnet_oracle as O


@pytest.fixture()
def emu(monkeypatch):
    emu_ops.install(monkeypatch)


def _model(b=2, h=64, w=128, lazy=True):
    from dcs_amd.trainer import TrainStep, make_opts
    ts = TrainStep(make_opts(criterion="focal", batch_size=b, lazy_pred_segmap=lazy), class_weight=None, device="cpu")
    ts.model.load_state_dict(O.make_state(seed=1), strict=True)
    img, labels, ldw, weather, cw = O.synthetic_batch(b, h, w, seed=5, two_crops=False, cell=16)
    ts.criterion.weight = cw
    return ts, img, labels, ldw


def test_training_forward_returns_a_handle_that_behaves_like_the_tensor(emu):
    from dcs_amd.losses import LazyLogits
    ts, img, labels, ldw = _model()
    seg, before, ff, ff0 = ts.model(img)
    assert isinstance(seg, LazyLogits) and seg._dense is None
    assert tuple(seg.shape) == (2, 19, 64, 128) and seg.dim() == 4 and seg.dtype == torch.float32 and seg.size(1) == 19
    assert seg._dense is None                                    # metadata queries do not materialise
    ts2, *_ = _model(lazy=False)
    seg_e = ts2.model(img)[0]
    assert not isinstance(seg_e, LazyLogits)
    am = seg.detach().max(dim=1)[1]                              # trainer.py:349-style use: materialises, same values
    assert seg._dense is not None and torch.equal(am, seg_e.detach().max(dim=1)[1])
    assert torch.allclose(seg[:, :, ::4, ::4], seg_e[:, :, ::4, ::4], rtol=0, atol=1e-6)
    ts.model.eval()
    with torch.no_grad():
        assert not isinstance(ts.model(img)[0], LazyLogits)      # eval / no-grad forwards stay eager


@pytest.mark.parametrize("crit", ["focal", "ce", "torch_ce"])
def test_fused_loss_equals_the_eager_path_and_backpropagates(emu, crit):
    from dcs_amd.losses import SemsegCrossEntropy
    res = []
    for lazy in (True, False):
        ts, img, labels, ldw = _model(lazy=lazy)
        seg, before, ff, ff0 = ts.model(img)
        lab = labels.clone()
        if crit == "focal":
            loss = ts.criterion(seg, lab, dict(label_distance_weight=ldw))
        elif crit == "ce":
            loss = SemsegCrossEntropy(ignore_id=255)(seg, lab)
        else:                                                    # the reference's ce_criterion: torch's own module
            loss = torch.nn.CrossEntropyLoss(ignore_index=255)(seg, lab)
        if lazy:
            assert (seg._dense is None) == (crit != "torch_ce")  # only a foreign consumer materialises
        loss.backward()
        res.append((float(loss), {k: p.grad.clone() for k, p in ts.model.named_parameters() if p.grad is not None}, lab))
    (l0, g0, lab0), (l1, g1, lab1) = res
    assert abs(l0 - l1) <= 1e-6 * abs(l1)
    assert torch.equal(lab0, lab1)                               # same in-place 255 -> 0 rewrite
    assert g0.keys() == g1.keys()
    for k in g0:
        assert float((g0[k] - g1[k]).norm()) <= 2e-4 * float(g1[k].norm()) + 1e-9, k
