"""Validate-side row: Evaluator against the golden from the reference's own Evaluator.
CPU: host path + device path with the emulated kernel.  GPU: the fused argmax + confusion-matrix kernel, from
full-resolution logits and from low-resolution logits upsampled on the fly (integer counts: bit-exact)."""
import os

import numpy as np
import pytest
import torch

import emu_ops


def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "metrics_evaluator.npz"), allow_pickle=False)


def logits_for(pred, C):
    """Logits whose argmax is `pred` (margin 1.0 plus noise < 0.4)."""
    g = np.random.default_rng(5)
    x = g.random((pred.shape[0], C) + pred.shape[1:]).astype(np.float32) * 0.4
    np.put_along_axis(x, pred[:, None].astype(np.int64), 1.5, axis=1)
    return torch.from_numpy(x)


def check_scores(ev, g):
    assert np.array_equal(ev.confusion_matrix, g["confusion"])
    assert np.array_equal(ev.confusion_matrix_sem_weather["0"], g["conf_w0"])
    assert np.array_equal(ev.confusion_matrix_sem_weather["2"], g["conf_w2"])
    assert abs(ev.Mean_Intersection_over_Union() - float(g["miou"])) < 1e-12
    assert abs(ev.Pixel_Accuracy() - float(g["acc"])) < 1e-12
    assert abs(ev.Pixel_Accuracy_Class() - float(g["acc_cls"])) < 1e-12
    assert abs(ev.Frequency_Weighted_Intersection_over_Union() - float(g["fwiou"])) < 1e-12
    pw = ev.Mean_Intersection_over_Union_each_weather()
    assert abs(pw["0"] - float(g["miou_w0"])) < 1e-10 and abs(pw["2"] - float(g["miou_w2"])) < 1e-10


def test_evaluator_host_path_matches_reference(golden_dir):
    from dcs_amd.metrics import Evaluator
    g = golden(golden_dir)
    ev = Evaluator(19, 4)
    ev.add_batch(g["gt"].astype(np.int64), g["pred"].astype(np.int64), g["weather"])
    check_scores(ev, g)


def test_evaluator_device_path_emulated(golden_dir, monkeypatch):
    emu_ops.install(monkeypatch)
    from dcs_amd.metrics import Evaluator
    g = golden(golden_dir)
    ev = Evaluator(19, 4)
    ev.add_batch_device(torch.from_numpy(g["gt"].astype(np.int64)), logits_for(g["pred"], 19), torch.from_numpy(g["weather"]))
    check_scores(ev, g)


@pytest.mark.gpu
def test_evaluator_device_path_gpu(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dcs_amd.metrics import Evaluator
    import dcs_amd.ops as ops
    g = golden(golden_dir)
    dev = "cuda:0"
    ev = Evaluator(19, 4)
    labels = torch.from_numpy(g["gt"].astype(np.int64)).to(dev)
    lg = logits_for(g["pred"], 19).to(dev)
    ev.add_batch_device(labels, lg, torch.from_numpy(g["weather"]).to(dev))
    check_scores(ev, g)
    # fused low-resolution path == argmax of the materialised upsample (same interpolation arithmetic)
    gen = np.random.default_rng(7)
    low = torch.zeros(2, 30, 50, 20)
    low[..., :19] = torch.from_numpy(gen.standard_normal((2, 30, 50, 19)).astype(np.float32))
    lab = torch.from_numpy(gen.integers(0, 19, size=(2, 120, 200)).astype(np.int64))
    lab[:, :3] = 255
    full = ops.upsample_to_nchw(low.to(dev), 19, 120, 200)
    c1 = torch.zeros(2, 19, 19, dtype=torch.int64, device=dev)
    c2 = torch.zeros(2, 19, 19, dtype=torch.int64, device=dev)
    p1 = ops.confusion(full, lab.to(dev), 19, c1, want_pred=True)
    p2 = ops.confusion(low.to(dev), lab.to(dev), 19, c2, lowres=(120, 200), want_pred=True)
    assert torch.equal(c1, c2) and torch.equal(p1, p2)
    assert torch.equal(p1.cpu().long(), full.cpu().argmax(1))
    assert int(c1.sum()) == int((lab != 255).sum())
    # through the Evaluator with the model's channels_last low-resolution view
    ev2 = Evaluator(19, 4)
    ev2.add_batch_device(lab.to(dev), low.to(dev)[..., :19].permute(0, 3, 1, 2), None, lowres=True)
    assert np.array_equal(ev2.confusion_matrix, c1.sum(0).cpu().numpy().astype(np.float64))
