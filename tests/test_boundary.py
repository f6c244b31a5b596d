"""Boundary-aware label weights: LabelBoundaryTransform of the reference
(dataloaders/custom_transforms_acdc.py:656-693) on the device.

PARITY UNPINNED against cv2 (not importable here, no fixture in the reference): the oracle restates OpenCV's published
3x3 chamfer transform and is pinned to the metric's definition by brute force; the HIP kernel (one transform for all
classes) is then compared BIT-EXACTLY (fixed-point distances) with the oracle's literal per-class statement."""
import numpy as np
import pytest
import torch

import emu_ops
from oracle import boundary_oracle as BO

DEV = "cuda:0"


def blocky_labels(B, H, W, seed, cell=9, ignore_frac=0.1, classes=19):
    g = np.random.default_rng(seed)
    out = np.empty((B, H, W), dtype=np.int64)
    for b in range(B):
        cy, cx = -(-H // cell), -(-W // cell)
        cells = g.integers(0, classes, size=(cy, cx))
        cells[g.random((cy, cx)) < ignore_frac] = 255
        out[b] = np.kron(cells, np.ones((cell, cell), dtype=np.int64))[:H, :W]
        out[b, : max(1, H // 16)] = 255                      # an ignore strip like the car hood / border
        noise = g.random((H, W)) < 0.01
        out[b][noise] = g.integers(0, classes, size=int(noise.sum()))
    return out


@pytest.mark.parametrize("H,W,seed", [(7, 9, 0), (12, 17, 1), (5, 31, 2), (16, 16, 3)])
def test_chamfer_restatement_equals_the_metric(H, W, seed):
    g = np.random.default_rng(seed)
    for p_zero in (0.05, 0.3, 0.9):
        mask = (g.random((H, W)) > p_zero).astype(np.uint8)
        assert np.array_equal(BO.chamfer3x3(mask), BO.brute_force(mask))
    ones = np.ones((H, W), np.uint8)                             # no zero pixel: everything saturates at DIST_MAX
    assert np.all(BO.chamfer3x3(ones) == np.float32(8192.0))
    one_zero = ones.copy(); one_zero[H // 2, W // 3] = 0
    assert np.array_equal(BO.chamfer3x3(one_zero), BO.brute_force(one_zero))


def test_transform_properties():
    lab = blocky_labels(1, 40, 56, 4)[0]
    w = BO.label_boundary_transform(lab, 19, True, 255)
    assert w.dtype == np.float32 and w.shape == lab.shape
    assert np.all(w[lab == 255] == 0) and np.all(w[lab != 255] > 0) and np.all(w <= 1)
    d = BO.label_boundary_transform(lab, 19, False, 255)
    assert d.shape == (19,) + lab.shape and np.all(d[:, lab == 255] == -1)
    # nearest differently-labelled pixel of a boundary pixel is one step away
    edge = (lab[1:, :] != lab[:-1, :]) & (lab[1:, :] != 255)
    assert np.all(d.max(0)[1:, :][edge] == np.float32(BO.HV / 65536.0))
    empty = np.full((8, 8), 255)                                  # lost-and-found style image without any label
    assert np.all(BO.label_boundary_transform(empty, 19, True, 255) == 0)
    uniform = np.full((8, 8), 3)                                  # one class only: distances saturate, std = 0 -> 1
    assert np.allclose(BO.label_boundary_transform(uniform, 19, True, 255), np.exp(-4096.0))


def test_host_class_cpu_emulated(monkeypatch):
    emu_ops.install(monkeypatch)
    from dcs_amd.boundary import LabelBoundaryTransform
    lab = blocky_labels(2, 24, 40, 5)
    ex = LabelBoundaryTransform(19, reduce=True)({"label": torch.from_numpy(lab)})
    for b in range(2):
        assert np.allclose(ex["label_distance_weight"][b].numpy(), BO.label_boundary_transform(lab[b], 19), atol=1e-6)
    one = LabelBoundaryTransform(19, reduce=True)({"label": torch.from_numpy(lab[0])})
    assert one["label_distance_weight"].shape == lab[0].shape
    full = LabelBoundaryTransform(19, reduce=False)({"label": torch.from_numpy(lab[1])})["label_distance_transform"]
    assert np.allclose(full.numpy(), BO.label_boundary_transform(lab[1], 19, False), atol=1e-6)


def test_refuses_cpu_tensors_without_emulation():
    from dcs_amd.boundary import LabelBoundaryTransform
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LabelBoundaryTransform(19, reduce=True)({"label": torch.zeros((4, 4), dtype=torch.int64)})


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,seed", [(2, 64, 96, 6), (1, 37, 53, 7), (3, 16, 1030, 8), (1, 130, 2049, 9), (1, 1, 1, 10),
                                        (1, 3, 4100, 11)])
def test_kernel_bit_exact_vs_oracle(B, H, W, seed):
    import dcs_amd.ops as ops
    lab = blocky_labels(B, H, W, seed)
    if W > 4096:
        with pytest.raises(RuntimeError):
            ops.label_boundary_weights(torch.from_numpy(lab).to(DEV), 19)
        return
    w, d = ops.label_boundary_weights(torch.from_numpy(lab).to(DEV), 19)
    torch.cuda.synchronize()
    for b in range(B):
        per = BO.label_boundary_transform(lab[b], 19, False)
        ref_d = np.maximum(per, 0).sum(0)                         # float32 distances, 0 at ignore pixels
        got_d = d[b].cpu().numpy()
        inside = lab[b] != 255
        got_f = got_d.astype(np.float32) * np.float32(1.0 / 65536.0)
        assert np.array_equal(got_f[inside], ref_d[inside]), "fixed-point distances must be bit-exact"
        ref_w = BO.label_boundary_transform(lab[b], 19, True)
        assert np.abs(w[b].cpu().numpy() - ref_w).max() < 2e-6


@pytest.mark.gpu
def test_kernel_degenerate_images():
    import dcs_amd.ops as ops
    lab = np.stack([np.full((20, 30), 255), np.full((20, 30), 4)]).astype(np.int64)
    lab[1, 10, 10] = 7
    w, d = ops.label_boundary_weights(torch.from_numpy(lab).to(DEV), 19)
    assert torch.all(w[0] == 0)
    assert np.abs(w[1].cpu().numpy() - BO.label_boundary_transform(lab[1], 19, True)).max() < 2e-6


@pytest.mark.gpu
def test_full_size_properties():
    """C3 label size: weights in [0,1], zero exactly on ignore pixels, one chamfer step at label boundaries,
    1-Lipschitz in the chamfer metric (neighbouring distances differ by at most one diagonal step)."""
    import dcs_amd.ops as ops
    lab = torch.from_numpy(blocky_labels(2, 1024, 2048, 12, cell=61)).to(DEV)
    w, d = ops.label_boundary_weights(lab, 19)
    assert bool(((w == 0) == (lab == 255)).all()) and float(w.max()) <= 1.0
    dd = d.to(torch.int64)
    assert int((dd[:, 1:, :] - dd[:, :-1, :]).abs().max()) <= BO.DIAG and int((dd[:, :, 1:] - dd[:, :, :-1]).abs().max()) <= BO.HV
    diff = lab[:, :, 1:] != lab[:, :, :-1]
    assert int(dd[:, :, 1:][diff].max()) == BO.HV and int(dd[:, :, :-1][diff].max()) == BO.HV
