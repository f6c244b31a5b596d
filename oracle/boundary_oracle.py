"""CPU restatement (numpy) of the reference's LabelBoundaryTransform
(dataloaders/custom_transforms_acdc.py:656-693).

TEST INFRASTRUCTURE: checker only (see oracle/__init__.py).

PARITY UNPINNED: the transform calls ``cv2.distanceTransform(mask, cv2.DIST_L2, maskSize=3)``; OpenCV (opencv-python,
un-pinned in the reference's requirements) is not importable in this image and the reference ships no fixture of
``label_distance_weight``.  ``chamfer3x3`` below restates OpenCV's published algorithm for that call
(modules/imgproc/src/distransform.cpp, ``distanceTransform_3x3``): two raster passes in 16.16 fixed point with
HV = cvRound(0.955 * 65536), DIAG = cvRound(1.3693 * 65536), a one-pixel border of DIST_MAX = INT_MAX >> 2, output
``(float)(t * 2^-16)``.  ``brute_force`` states the metric those passes compute (shortest 8-connected path to the nearest
zero pixel) and pins ``chamfer3x3`` on small cases in tests/test_boundary.py.
"""
from __future__ import annotations

import numpy as np

HV = int(round(0.955 * 65536))      # 62587
DIAG = int(round(1.3693 * 65536))   # 89738
DIST_MAX = (2 ** 31 - 1) >> 2


def chamfer3x3(mask: np.ndarray) -> np.ndarray:
    """mask uint8 [H,W]; distance (float32) of every non-zero pixel to the nearest zero pixel; 0 at zero pixels."""
    H, W = mask.shape
    tmp = np.full((H + 2, W + 2), DIST_MAX, dtype=np.int64)
    k = np.arange(W, dtype=np.int64)
    for y in range(H):                                   # forward pass
        up = tmp[y]                                      # row y-1 in padded coordinates
        t = np.minimum(np.minimum(up[0:W] + DIAG, up[1:W + 1] + HV), up[2:W + 2] + DIAG)
        t = np.where(mask[y] == 0, 0, t)
        # in-row recurrence t[x] = min(t[x], t[x-1] + HV) for non-zero pixels, zero pixels restart it at 0
        left = np.minimum(np.minimum.accumulate(np.concatenate(([DIST_MAX + HV], t[:-1] + HV)) - HV * k) + HV * k,
                          DIST_MAX + HV * (k + 1))
        tmp[y + 1, 1:W + 1] = np.where(mask[y] == 0, 0, np.minimum(t, left))
    out = np.empty((H, W), dtype=np.float32)
    for y in range(H - 1, -1, -1):                       # backward pass
        dn = tmp[y + 2]
        t0 = tmp[y + 1, 1:W + 1]
        t = np.minimum(np.minimum(np.minimum(dn[2:W + 2] + DIAG, dn[1:W + 1] + HV), dn[0:W] + DIAG), t0)
        tr = t[::-1]
        right = (np.minimum.accumulate(np.concatenate(([DIST_MAX + HV], tr[:-1] + HV)) - HV * k) + HV * k)
        t = np.minimum(tr, right)[::-1]
        t = np.where(t0 > HV, t, t0)                     # OpenCV only relaxes pixels with t0 > HV_DIST
        tmp[y + 1, 1:W + 1] = t
        out[y] = (np.minimum(t, DIST_MAX).astype(np.float32) * np.float32(1.0 / 65536.0))
    return out


def brute_force(mask: np.ndarray) -> np.ndarray:
    """Shortest 8-connected path (edge step HV, diagonal step DIAG) to the nearest zero pixel, by definition."""
    H, W = mask.shape
    zy, zx = np.nonzero(mask == 0)
    out = np.zeros((H, W), dtype=np.float32)
    for y in range(H):
        for x in range(W):
            if mask[y, x] == 0:
                continue
            if len(zy) == 0:
                d = DIST_MAX
            else:
                dy, dx = np.abs(zy - y), np.abs(zx - x)
                d = int((np.minimum(dy, dx) * DIAG + np.abs(dy - dx) * HV).min())
            out[y, x] = np.float32(min(d, DIST_MAX)) * np.float32(1.0 / 65536.0)
    return out


def label_boundary_transform(labels: np.ndarray, num_classes: int, reduce: bool = True, ignore_id: int = 255,
                             dist_fn=chamfer3x3):
    """custom_transforms_acdc.py:663-693 line by line (cv2.distanceTransform -> dist_fn)."""
    labels = np.array(labels)
    present = np.unique(labels)
    distances = np.zeros([num_classes] + list(labels.shape), dtype=np.float32) - 1.0
    for i in range(num_classes):
        if i not in present:
            continue
        class_mask = labels == i
        distances[i][class_mask] = dist_fn(np.uint8(class_mask))[class_mask]
    if not reduce:
        return distances
    ignore_mask = labels == ignore_id
    distances[distances < 0] = 0
    distances = distances.sum(axis=0)
    std_d = np.std(distances)
    if std_d == 0:
        std_d = 1
    distance_factor = distances / (2 * std_d)
    label_distances = np.exp(-distance_factor)
    label_distances[ignore_mask] = 0
    return label_distances
