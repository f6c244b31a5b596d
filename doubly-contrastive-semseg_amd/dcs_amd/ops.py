"""Tensor-level wrappers over the C ABI (include/dcs_hip.h).

PyTorch supplies device memory and the current HIP stream only; every
computation below is a kernel of libdcs_hip.so.  Activations are 4-D
``[N, H, W, C]`` contiguous fp32 tensors (NHWC), convolution weights are the
reference's OIHW parameters held in channels_last memory format (= KRSC).
All functions require CUDA(HIP) tensors; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import lib as _lib
from .lib import DcsConvGeom

_F32 = torch.float32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream of the current device as a raw handle.  torch.cuda.current_stream() builds a Python Stream
    object per call (~8 us; 1100 launches per step: the small configurations are host-bound); the two C entry points
    behind it cost ~0.5 us."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if _batch is not None:
        _batch.keep.append(t)          # a recorded launch reads / writes it later: its memory must not be recycled before
    return C.c_void_p(t.data_ptr())


_fn_cache = {}


def _call_now(name: str, *args):
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(_lib.load(), name)
    rc = fn(*args)
    if rc != 0:
        _lib.check(rc, name)


def _call(name: str, *args):
    if _batch is None:
        _call_now(name, *args)
    else:
        _batch.record(name, args)


# --------------------------------------------------------------------------- #
# level batching: the pyramid levels of one layer as ONE launch per convolution
# --------------------------------------------------------------------------- #
# The three pyramid levels apply the same modules to maps of different size (network/backbone/resnet_pyramid.py:318-341).
# Inside ``with level_batch() as lb`` the model runs a block for level 0, 1, 2 in turn (lb.level(i) before each) and then
# lb.flush(): launches are RECORDED per level and emitted as a merge of the per-level sequences that keeps every level's
# own order, in which the convolution launches standing at the head of several levels go out through the *_multi entries
# (one grid, bitwise the per-level results: include/dcs_hip.h).  Launches of different levels that update the same
# memory (running statistics, the shared weight / BatchNorm parameter gradients) keep their recorded order.
_BATCHABLE = {"dcs_conv_gather_x3": ("dcs_conv_gather_x3_multi", _lib.DcsGatherLaunch),
              "dcs_conv3x3_x3w": ("dcs_conv3x3_x3w_multi", _lib.DcsGatherLaunch),
              "dcs_conv_wgrad_x3": ("dcs_conv_wgrad_x3_multi", _lib.DcsWgradLaunch)}
# entry -> index of the argument that names memory shared between the levels (None argument: nothing shared)
_ORDER_KEY_ARG = {"dcs_reduce_slab": 1, "dcs_bn_finalize": 3, "dcs_bn_ema_again": 1, "dcs_bn_bwd_apply": 8,
                  "dcs_bn_pool_bwd_apply": 7}
_batch = None          # the active recorder (one per process: the step runs on one host thread)
launch_counts = {"single": 0, "multi": 0, "merged": 0}      # convolution launches emitted by level batches (diagnostics)


def _ptr(a):
    return a.value if isinstance(a, C.c_void_p) else a


def _gather_struct(a, x3w):
    if x3w:      # dcs_conv3x3_x3w(src, wfrag, bias, dst, geom, accumulate, stats, pro, bn_y, bn_mask, bn, relu, src_max, stream)
        src, w, bias, dst, g, acc, stats, pro, yb, mb, bnr, relu, smax, _ = a
        ns, slab = 1, 0
    else:        # dcs_conv_gather_x3(..., relu, nsplit, slab_stride, src_max, stream)
        src, w, bias, dst, g, acc, stats, pro, yb, mb, bnr, relu, ns, slab, smax, _ = a
    return _lib.DcsGatherLaunch(_ptr(src), _ptr(w), _ptr(bias), _ptr(dst), C.pointer(g), _ptr(stats), _ptr(pro), _ptr(yb),
                                _ptr(mb), _ptr(bnr), slab, acc, relu, ns, _ptr(smax))


def _wgrad_struct(a):
    src, dy, slab, g, dcs, split0, ns, pro, dmax, _ = a      # dcs_conv_wgrad_x3(src, dy, slab, geom, dy_cstride, split0, nsplit, pro, dy_max, stream)
    return _lib.DcsWgradLaunch(_ptr(src), _ptr(dy), _ptr(slab), C.pointer(g), _ptr(pro), dcs, split0, ns, _ptr(dmax))


class _LevelBatch:
    def __init__(self):
        self.seqs: List[list] = []
        self.cur = 0
        self.keep: list = []
        self.keyq = {}
        self.n = 0

    def record(self, name, args):
        while len(self.seqs) <= self.cur:
            self.seqs.append([])
        key = None
        ka = _ORDER_KEY_ARG.get(name)
        if ka is not None and args[ka] is not None:
            key = _ptr(args[ka])            # the memory itself: different entries updating it are ordered too
            self.keyq.setdefault(key, []).append(self.n)
        self.seqs[self.cur].append((name, args, key, self.n))
        self.n += 1

    def emit(self):
        seqs = [q for q in self.seqs if q]
        at = [0] * len(seqs)
        kpos = {k: 0 for k in self.keyq}
        left = sum(len(q) for q in seqs)
        while left:
            progress = False
            moved = True
            while moved:               # every level up to its next convolution (until nothing moves: a level may wait for
                moved = False          # another level's earlier update of shared memory)
                for L, q in enumerate(seqs):
                    while at[L] < len(q) and q[at[L]][0] not in _BATCHABLE:
                        name, args, key, cid = q[at[L]]
                        if key is not None:
                            if self.keyq[key][kpos[key]] != cid:
                                break                  # an earlier update of the same memory by another level comes first
                            kpos[key] += 1
                        if name == "<py>":
                            args()
                        else:
                            _call_now(name, *args)
                        at[L] += 1
                        left -= 1
                        moved = progress = True
            heads = [L for L, q in enumerate(seqs) if at[L] < len(q) and q[at[L]][0] in _BATCHABLE]
            if heads:
                names = [seqs[L][at[L]][0] for L in heads]
                name = max(names, key=names.count)              # the entry most levels are waiting at (ties: the first)
                grp = [L for L in heads if seqs[L][at[L]][0] == name][:_lib.MULTI_MAX]
                calls = [seqs[L][at[L]][1] for L in grp]
                if len(grp) == 1:
                    _call_now(name, *calls[0])
                    launch_counts["single"] += 1
                else:
                    multi, struct = _BATCHABLE[name]
                    if struct is _lib.DcsWgradLaunch:
                        arr = (struct * len(grp))(*[_wgrad_struct(a) for a in calls])
                    else:
                        arr = (struct * len(grp))(*[_gather_struct(a, name == "dcs_conv3x3_x3w") for a in calls])
                    _call_now(multi, arr, len(grp), calls[0][-1])
                    launch_counts["multi"] += 1
                    launch_counts["merged"] += len(grp)
                for L in grp:
                    at[L] += 1
                left -= len(grp)
                progress = True
            if not progress:                                  # cannot happen (see the ordering argument in DESIGN.md)
                raise RuntimeError("level batch: no launch is ready")


class level_batch:
    """Context manager, see above.  DCS_LEVEL_BATCH=0 (or a nested use) turns it into a no-op: every launch then goes out
    at once, level after level."""

    def __enter__(self):
        global _batch
        self.own = _batch is None and os.environ.get("DCS_LEVEL_BATCH", "1") != "0"
        if self.own:
            _batch = _LevelBatch()
        return self

    def level(self, i: int):
        if self.own:
            _batch.cur = i

    def flush(self):
        global _batch
        if not self.own or _batch is None:
            return
        b, _batch = _batch, None                 # emission itself launches at once
        try:
            b.emit()
        finally:
            _batch = _LevelBatch()

    def __exit__(self, et, ev, tb):
        global _batch
        if self.own:
            b, _batch = _batch, None
            if et is None:
                b.emit()
        return False


def _defer(fn):
    """A torch operation on data that recorded launches produce or consume: runs in its place of the level's sequence."""
    if _batch is None:
        fn()
    else:
        _batch.record("<py>", fn)


def _req(t: torch.Tensor, dtype=_F32):
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"dcs_amd.ops needs contiguous {dtype} device tensors, got {t.device} {t.dtype} "
                           f"contiguous={t.is_contiguous()} (no CPU fallback)")
    return t


def require_device(t: torch.Tensor, what="input"):
    """The product path runs on the GPU only; a CPU tensor is an error, never a fallback."""
    if not t.is_cuda:
        raise RuntimeError(f"dcs_amd runs on MI355X only (no CPU path): {what} is a CPU tensor")


def krsc(w: torch.Tensor) -> torch.Tensor:
    """[Cout,R,S,Cin] contiguous view of an OIHW channels_last parameter (no copy)."""
    v = w.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        raise RuntimeError("convolution weights must be in channels_last memory format")
    return v


# --------------------------------------------------------------------------- #
# geometry (cached per shape)
# --------------------------------------------------------------------------- #
_geom_cache = {}


def _mk_geom(N, SH, SW, DH, DW, TY, TX, sy, dsy, dy0, dx0, K, Cout, taps, wstride, src_cs, dst_cs, stem=0):
    g = DcsConvGeom()
    g.N, g.SH, g.SW, g.DH, g.DW, g.TY, g.TX = N, SH, SW, DH, DW, TY, TX
    g.sy = g.sx = sy
    g.dsy = g.dsx = dsy
    g.dy0, g.dx0 = dy0, dx0
    g.K, g.Cout, g.ntaps, g.wstride = K, Cout, len(taps), wstride
    g.src_cstride, g.dst_cstride, g.stem = src_cs, dst_cs, stem
    for i, (oy, ox, wo) in enumerate(taps):
        g.offy[i], g.offx[i], g.wofs[i] = oy, ox, wo
    return g


def out_size(n, k, stride, pad):
    return (n + 2 * pad - k) // stride + 1


def out_size_d(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


def geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad, src_cs=None, dst_cs=None, dil=1, koff=0, ktot=None):
    """Forward gather geometry.  dil: dilation.  (koff, ktot): the Cin source channels multiply the weight channel
    slice [koff, koff+Cin) of rows that hold ktot channels per tap ("virtual concat" along the input channels)."""
    ktot = ktot or Cin
    key = ("f", N, H, W, Cin, Cout, R, S, stride, pad, src_cs, dst_cs, dil, koff, ktot)
    g = _geom_cache.get(key)
    if g is None:
        OH, OW = out_size_d(H, R, stride, pad, dil), out_size_d(W, S, stride, pad, dil)
        taps = [(r * dil - pad, s * dil - pad, (r * S + s) * ktot + koff) for r in range(R) for s in range(S)]
        g = _mk_geom(N, H, W, OH, OW, OH, OW, stride, 1, 0, 0, Cin, Cout, taps, R * S * ktot,
                     src_cs or Cin, dst_cs or Cout)
        _geom_cache[key] = g
    return g


def geoms_dgrad(N, IH, IW, Cin, Cout, R, S, stride, pad, dil=1):
    """One geometry per input parity class (no wasted taps for stride 2)."""
    key = ("d", N, IH, IW, Cin, Cout, R, S, stride, pad, dil)
    gs = _geom_cache.get(key)
    if gs is None:
        OH, OW = out_size_d(IH, R, stride, pad, dil), out_size_d(IW, S, stride, pad, dil)
        gs = []
        for py in range(stride):
            for px in range(stride):
                TY, TX = -(-(IH - py) // stride), -(-(IW - px) // stride)
                if TY <= 0 or TX <= 0:
                    continue
                taps = [((py + pad - r * dil) // stride, (px + pad - s * dil) // stride, (r * S + s) * Cout)
                        for r in range(R) for s in range(S)
                        if (py + pad - r * dil) % stride == 0 and (px + pad - s * dil) % stride == 0]
                if not taps:
                    gs.append(None)
                    continue
                gs.append(_mk_geom(N, OH, OW, IH, IW, TY, TX, 1, stride, py, px, Cout, Cin, taps, R * S * Cout,
                                   Cout, Cin))
        _geom_cache[key] = gs
    return gs


def geom_stem_fwd(N, H, W):
    """Forward-only variant: every filter row as two half rows of 4 pixels (14 taps of 16 floats; the 8th pixel slot of
    the packed weight row is zero, so its data never matters)."""
    key = ("s14", N, H, W)
    g = _geom_cache.get(key)
    if g is None:
        OH, OW = out_size(H, 7, 2, 3), out_size(W, 7, 2, 3)
        taps = [(r - 3, -3 + 4 * c, r * 32 + 16 * c) for r in range(7) for c in range(2)]
        g = _mk_geom(N, H, W, OH, OW, OH, OW, 2, 1, 0, 0, 4, 64, taps, 224, 4, 64, stem=1)
        _geom_cache[key] = g
    return g


def geom_stem(N, H, W):
    key = ("s", N, H, W)
    g = _geom_cache.get(key)
    if g is None:
        OH, OW = out_size(H, 7, 2, 3), out_size(W, 7, 2, 3)
        taps = [(r - 3, -3, r * 32) for r in range(7)]
        g = _mk_geom(N, H, W, OH, OW, OH, OW, 2, 1, 0, 0, 4, 64, taps, 224, 4, 64, stem=1)
        _geom_cache[key] = g
    return g


# --------------------------------------------------------------------------- #
# convolution
# --------------------------------------------------------------------------- #
CONV_BM = 128
PRO_MAXK = 512            # DCS_PRO_MAXK of include/dcs_hip.h


def _stats_buffer(M, Cout, dev):
    """Per-row-tile partial sums written by the conv epilogue, padded so they reduce in <= 2 fixed-order levels."""
    G = -(-M // CONV_BM)
    if G <= 2048:
        return torch.empty((G, 2, Cout), device=dev, dtype=_F32), G, 0
    G1 = -(-G // 128)
    return torch.zeros((G1 * 128, 2, Cout), device=dev, dtype=_F32), G, G1


class Moments(torch.Tensor):
    """[1,2,C] tensor holding (mean, biased variance) instead of (sum, sum of squares): see bn_finalize."""


def _stats_reduce(part, G, G1, Cout, count):
    """-> Moments [1, 2, Cout] = (mean, biased variance) over all ``count`` rows, formed in double (deterministic)."""
    out = torch.empty((1, 2, Cout), device=part.device, dtype=_F32)
    if G1 == 0:
        _call("dcs_colsum_final", _p(part), _p(out), 1, G, Cout, 1.0, float(count), _stream())
    else:
        mid = torch.empty((G1, 2, Cout), device=part.device, dtype=_F32)
        _call("dcs_colsum_final", _p(part), _p(mid), G1, 128, Cout, 1.0, 0.0, _stream())
        _call("dcs_colsum_final", _p(mid), _p(out), 1, G1, Cout, 1.0, float(count), _stream())
    return out.as_subclass(Moments)


def _ksplit(g, M, Cout):
    """Number of K splits for a gather launch: 1 unless the launch has too few output tiles to fill the chip AND a long
    reduction (deep layers of small inputs, e.g. 512-channel 3x3 on 8x16 maps: 128 blocks x 144 chunks)."""
    if g.stem or g.dsy != 1 or g.dst_cstride != Cout or os.environ.get("DCS_KSPLIT", "1") == "0":
        return 1
    bn = 128 if Cout > 64 else (64 if Cout > 32 else 32)
    blocks = (-(-M // CONV_BM)) * (-(-Cout // bn))
    nch32 = g.ntaps * (-(-g.K // 32))
    if blocks > 160 or nch32 < 24:
        return 1
    # e.g. 512 -> 512 channels 3x3 on 8 images of 4x8 pixels: 8 blocks x 144 chunks = 300 us serial; 16 splits: 25 us
    return int(max(1, min(16, 512 // blocks, nch32 // 6)))


def x3_ok(g):
    """Does the launch run on the split-bf16 kernels (csrc/conv_split.hip: fp32 operands as three bf16 pieces, six bf16
    MFMAs per product, fp32-class error at 6/16 of the fp32-MFMA time)?  DCS_CONV_X3=0 keeps the exact fp32 MFMA."""
    ok = getattr(g, "_x3", None)
    if ok is None:
        if g.stem:                               # the 14-tap stem geometry: a tap = one 16-float chunk
            ok = g.ntaps == 14 and g.Cout == 64 and g.wstride == 224
        else:
            ok = g.Cout > 32 and g.K % 16 == 0 and g.wstride % 16 == 0
        ok = ok and all(g.wofs[t] % 16 == 0 for t in range(g.ntaps))
        g._x3 = ok
    if g.stem and os.environ.get("DCS_STEM_X3", "1") == "0":
        return False
    return ok and os.environ.get("DCS_CONV_X3", "1") != "0"


_split_cache = {}


_max_pool = {"buf": None, "at": 0}       # device words for per-tensor maxima (bn_bwd -> the fp16 two-piece kernels)


def _max_slot(device):
    """One zeroed uint32 device word (a view into a pool that is re-zeroed once per step / when exhausted)."""
    mp = _max_pool
    if mp["buf"] is None or mp["buf"].device != device or mp["at"] >= mp["buf"].numel():
        mp["buf"] = torch.zeros((1024,), device=device, dtype=torch.int32)
        mp["at"] = 0
    s = mp["buf"][mp["at"]:mp["at"] + 1]
    mp["at"] += 1
    return s


def tag_max(t):
    """Attach the device word with max |t| to a gradient tensor that no bn_bwd produced (one read of the tensor), so that
    the 3x3 convolutions consuming it -- data and weight gradient -- take the fp16 two-piece kernels.  The caller
    guarantees that t is not written again before they ran.  No-op on the CPU / with DCS_X2H=0 / when already tagged."""
    if t is None or not t.is_cuda or not x2h_on() or hasattr(t, "_dcs_max") or t.numel() % 4 or not t.is_contiguous():
        return t
    slot = _max_slot(t.device)
    _call("dcs_maxabs", _p(t), t.numel(), _p(slot), _stream())
    t._dcs_max = slot
    return t


def split_weight(wk):
    """[rows, ...] fp32 (row length % 16 == 0) -> the three-piece bf16 image dcs_conv_gather_x3 stages (6 B / element).
    Cached per step by storage address; the cache entry keeps ``wk`` alive, so the address cannot be recycled."""
    key = (wk.data_ptr(), tuple(wk.shape), tuple(wk.stride()))
    hit = _split_cache.get(key)
    if hit is not None:
        return hit[1]
    rows = wk.shape[0]
    ws = wk.numel() // rows
    out = torch.empty((rows, ws * 3 // 2), device=wk.device, dtype=_F32)
    _call_now("dcs_split_weight", _p(wk), _p(out), rows, ws, _stream())
    if len(_split_cache) >= 1024:            # callers that never run a model forward (micro-benchmarks, tests)
        _split_cache.clear()
    _split_cache[key] = (wk, out)
    return out


def split_weight_frag(wk):
    """[rows, ...] fp32 -> the fragment-major split image (+ sign-flipped copy) dcs_conv3x3_x3w reads straight into its MFMA
    operands; cached per step like split_weight."""
    key = ("frag", wk.data_ptr(), tuple(wk.shape), tuple(wk.stride()))
    hit = _split_cache.get(key)
    if hit is not None:
        return hit[1]
    rows = wk.shape[0]
    ws = wk.numel() // rows
    J = -(-rows // 32)
    out = torch.empty((2 * (ws // 16) * J * 3 * 256,), device=wk.device, dtype=_F32)
    _call_now("dcs_split_weight_frag", _p(wk), _p(out), rows, ws, _stream())
    if len(_split_cache) >= 1024:
        _split_cache.clear()
    _split_cache[key] = (wk, out)
    return out


ACC_FP16X2 = 16
ACC_WFRAG = 32         # DCS_ACC_WFRAG: dcs_conv_gather_x3 reads a fragment-major weight image          # DCS_ACC_FP16X2 (include/dcs_hip.h)


def split_weight_h2(wk):
    """The fp16 two-piece form of split_weight (4 B / element), cached per step."""
    key = ("h2", wk.data_ptr(), tuple(wk.shape), tuple(wk.stride()))
    hit = _split_cache.get(key)
    if hit is not None:
        return hit[1]
    rows = wk.shape[0]
    ws = wk.numel() // rows
    out = torch.empty((rows * ws,), device=wk.device, dtype=_F32)
    _call_now("dcs_split_weight_h2", _p(wk), _p(out), rows, ws, _stream())
    if len(_split_cache) >= 1024:
        _split_cache.clear()
    _split_cache[key] = (wk, out)
    return out


def split_weight_frag_h2(wk):
    """The fp16 two-piece form of split_weight_frag (forward 3x3 convolutions, three MFMAs per product): cached per step."""
    key = ("frag_h2", wk.data_ptr(), tuple(wk.shape), tuple(wk.stride()))
    hit = _split_cache.get(key)
    if hit is not None:
        return hit[1]
    rows = wk.shape[0]
    ws = wk.numel() // rows
    J = -(-rows // 32)
    out = torch.empty((2 * (ws // 16) * J * 2 * 256,), device=wk.device, dtype=_F32)
    _call_now("dcs_split_weight_frag_h2", _p(wk), _p(out), rows, ws, _stream())
    if len(_split_cache) >= 1024:
        _split_cache.clear()
    _split_cache[key] = (wk, out)
    return out


def x2h_on():
    """Convolutions on two fp16 pieces per operand where they apply (DCS_X2H=0: three bf16 pieces everywhere)."""
    return os.environ.get("DCS_X2H", "1") != "0"


# Forward convolutions take the fp16 two-piece kernels only while the activations are known to be of order one: in
# TRAINING mode every convolution input is a batch-normalised tensor (or a short sum of them, or the normalised image), far
# inside the kernels' domain |x| < 16384.  An eval-mode forward normalises with running statistics that need not match the
# data (a freshly initialised ResNet-101 lets its activations grow through 33 residual blocks): it keeps the three-piece
# bf16 kernels, whose range is fp32's.  Gradients are always scaled by their measured maximum.
_x2h_forward = [True]
_step_max_m = [0]                  # pixels of the largest convolution output since new_step (conv_wgrad's split policy)


def new_step(training=True):
    """Start of a model forward.  Weights may change between steps (optimizer): forget the split images of the previous
    step; within one forward + backward a weight (or its data-gradient repack) is split once and reused by the three
    pyramid levels.  training: batch-statistics BatchNorm -> forward convolutions may use the fp16 two-piece kernels."""
    _split_cache.clear()
    _max_pool["buf"] = None
    _x2h_forward[0] = bool(training)
    _step_max_m[0] = 0


def x3w_ok(g):
    """Dense 3x3 / stride 1 launches (8 x 32-pixel tiles of <= 64 channels, 4 x 32 of 128) with at least 256 tiles run
    the halo-resident kernel that takes its weight fragments straight from global memory (csrc/conv_split.hip,
    conv3x3_x3w_kernel: bitwise the results of the LDS-staged weights, +4..13 % on 64 channels, +5..11 % on 128..512;
    against the per-tap kernel on small maps: 16x32 of 512 channels 169 -> 196 TF, 32x64 of 128 channels 147 -> 168)."""
    ok = getattr(g, "_x3w", None)
    if ok is None:
        offs = {(g.offy[t], g.offx[t]) for t in range(g.ntaps)}
        ok = (not g.stem and g.ntaps == 9 and g.sy == 1 and g.dsy == 1 and g.dy0 == 0 and g.dx0 == 0 and g.Cout > 32 and
              g.K % 32 == 0 and g.wstride % 16 == 0 and g.TX % 32 == 0 and g.TY % (8 if g.Cout <= 64 else 4) == 0 and
              g.SH == g.TY and g.SW == g.TX and g.DH == g.TY and g.DW == g.TX and all(g.wofs[t] % 16 == 0 for t in range(g.ntaps)) and
              offs == {(a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)} and
              (g.N * g.TY * g.TX // (256 if g.Cout <= 64 else 128)) * (-(-g.Cout // 128)) >=
              int(os.environ.get("DCS_X3W_MIN", "256")) and
              g.SH * g.SW * g.src_cstride * 4 < 2 ** 31)
        g._x3w = ok
    return ok and os.environ.get("DCS_X3W", "1") != "0" and os.environ.get("DCS_X3_HALO", "1") == "1"


def stem7_ok(g):
    """The 7x7 stem forward (14-tap form) on 8 x 32-pixel output tiles with at least 256 of them runs the kernel that keeps
    its input patch in LDS (csrc/conv_split.hip, stem7_h2_kernel; fp16 two-piece form).  DCS_STEM7=0: the per-tap kernel."""
    ok = getattr(g, "_stem7", None)
    if ok is None:
        ok = bool(g.stem and g.ntaps == 14 and g.Cout == 64 and g.wstride == 224 and g.TX % 32 == 0 and g.TY % 8 == 0 and
                  g.DH == g.TY and g.DW == g.TX and g.SH >= 2 * g.TY - 1 and g.SW >= 2 * g.TX - 1 and
                  g.N * g.TY * g.TX // 256 >= 256 and g.SH * g.SW * 16 < 2 ** 31)
        g._stem7 = ok
    return ok and os.environ.get("DCS_STEM7", "1") != "0"


def _gather_launch(src, wgt, bias, dst, g, accumulate, stats, pro=None, bnb=None, ns=1, slab_n=0, fwd=False):
    """One launch of the gather kernel family.  bnb = (y, mask, bn record, relu) or None.  fwd: a forward convolution
    (activations x weights: operands of known magnitude -> the fp16 two-piece kernel where it applies); a data gradient
    takes that kernel when its source carries the device word with its maximum (bn_bwd's dy: ``_dcs_max``)."""
    yb, mb, bnr, relu = bnb if bnb is not None else (None, None, None, False)
    m8 = getattr(mb, "_mask8", None) if (mb is not None and mask8_on()) else None
    if m8 is not None:                       # the byte mask dcs_bn_act left beside the tensor: relu = 2 (include/dcs_hip.h)
        # + 4: the launch stores the MASKED gradient (nothing reads a residual block's incoming gradient except through its
        # ReLU), so that bn_bwd(want_gm=True) can hand the tensor back as gm instead of writing it again
        if os.environ.get("DCS_STORE_MASKED", "1") != "0":
            dst._dcs_masked = mb
            relu = 6
        else:
            relu = 2
        mb = m8
    relu = int(relu)
    if (ns == 1 and g.stem and fwd and bias is None and pro is None and bnb is None and x3_ok(g) and stem7_ok(g) and
            _x2h_forward[0] and x2h_on()):
        _call("dcs_conv3x3_x3w", _p(src), _p(split_weight_frag_h2(wgt)), None, _p(dst), g, accumulate | ACC_FP16X2,
              _p(stats), None, None, None, None, 0, None, _stream())
    elif ns == 1 and x3_ok(g) and x3w_ok(g):
        smax = None if fwd else getattr(src, "_dcs_max", None)
        if ((fwd and _x2h_forward[0]) or smax is not None) and x2h_on():
            _call("dcs_conv3x3_x3w", _p(src), _p(split_weight_frag_h2(wgt)), _p(bias), _p(dst), g, accumulate | ACC_FP16X2,
                  _p(stats), _p(pro), _p(yb), _p(mb), _p(bnr), relu, _p(smax), _stream())
        else:
            _call("dcs_conv3x3_x3w", _p(src), _p(split_weight_frag(wgt)), _p(bias), _p(dst), g, accumulate, _p(stats),
                  _p(pro), _p(yb), _p(mb), _p(bnr), relu, None, _stream())
    elif x3_ok(g):
        smax = None if fwd else getattr(src, "_dcs_max", None)
        if ((fwd and _x2h_forward[0]) or smax is not None) and x2h_on():
            if not g.stem and os.environ.get("DCS_TAP_WFRAG", "1") != "0":
                # weight fragments straight from global memory (fragment-major image): the LDS-staged fp16 form is
                # LDS-bandwidth bound
                _call("dcs_conv_gather_x3", _p(src), _p(split_weight_frag_h2(wgt)), _p(bias), _p(dst), g,
                      accumulate | ACC_FP16X2 | ACC_WFRAG, _p(stats), _p(pro), _p(yb), _p(mb), _p(bnr), relu, ns,
                      slab_n, _p(smax), _stream())
            else:
                _call("dcs_conv_gather_x3", _p(src), _p(split_weight_h2(wgt)), _p(bias), _p(dst), g, accumulate | ACC_FP16X2,
                      _p(stats), _p(pro), _p(yb), _p(mb), _p(bnr), relu, ns, slab_n, _p(smax), _stream())
        else:
            _call("dcs_conv_gather_x3", _p(src), _p(split_weight(wgt)), _p(bias), _p(dst), g, accumulate, _p(stats),
                  _p(pro), _p(yb), _p(mb), _p(bnr), relu, ns, slab_n, None, _stream())
    elif bnb is not None:
        assert pro is None and ns == 1
        _call("dcs_conv_gather_bnbwd", _p(src), _p(wgt), _p(dst), C.byref(g), accumulate, _p(yb), _p(mb), _p(bnr),
              relu, _p(stats), _stream())
    elif pro is not None:
        _call("dcs_conv_gather_pro", _p(src), _p(wgt), _p(bias), _p(dst), C.byref(g), accumulate, _p(stats), _p(pro), ns,
              slab_n, _stream())
    elif ns > 1:
        _call("dcs_conv_gather_split", _p(src), _p(wgt), _p(dst), C.byref(g), ns, slab_n, _stream())
    else:
        _call("dcs_conv_gather", _p(src), _p(wgt), _p(bias), _p(dst), C.byref(g), accumulate, _p(stats), _stream())


def _gather_split(src, wgt, g, ns, dst, accumulate, pro=None):
    """Split-K launch + fixed-order slab reduce into dst (dense [N,DH,DW,Cout])."""
    n = dst.numel()
    slab = torch.empty((ns, n), device=dst.device, dtype=_F32)
    _gather_launch(src, wgt, None, slab, g, 0, None, pro, None, ns, n)
    _call("dcs_reduce_slab", _p(slab), _p(dst), n, ns, 1 if accumulate else 0, 0, 0, _stream())


def _gather(src, wgt, bias, dst, g, accumulate, stats, pro):
    _gather_launch(src, wgt, bias, dst, g, accumulate, stats, pro, fwd=True)


def pro_ok(Cin):
    """Can a convolution over Cin source channels apply the BatchNorm + ReLU of its input as a prologue?"""
    return Cin <= PRO_MAXK and os.environ.get("DCS_PROLOGUE", "1") != "0"


def conv_fwd(x, w, stride, pad, bias=None, dst_cs=None, want_stats=False, dil=1, koff=None, out=None, pro=None,
             stats_images=None):
    """nn.Conv2d forward.  x [N,H,W,Cin]; w OIHW channels_last; -> [N,OH,OW,dst_cs or Cout].
    want_stats: also return sums [1,2,Cout] (per-channel sum / sum of squares of the output) from the fused epilogue.
    koff: x holds the input-channel slice [koff, koff+Cin) of a wider weight (the other slices belong to other
    tensors of a concatenation); out: accumulate into this tensor instead of allocating (sum over the slices).
    pro: BatchNorm record [4,Cin] of x -- the convolution reads relu(x * scale + shift) (pro_ok(Cin) must hold).
    stats_images: the statistics cover the first stats_images images only (the segmentation head normalises the first
    crop of a two-crop batch, network/weathernet.py:78-82): a prefix of the epilogue's per-tile rows."""
    _req(x)
    N, H, W, Cin = x.shape
    Cout, Ktot, R, S = w.shape
    g = geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad, None, dst_cs, dil, koff or 0, Ktot)
    _step_max_m[0] = max(_step_max_m[0], N * g.DH * g.DW)
    cs = dst_cs or Cout
    ns = _ksplit(g, N * g.DH * g.DW, Cout) if bias is None else 1
    if out is not None:
        assert not want_stats
        if ns > 1 and out.is_contiguous():
            _gather_split(x, krsc(w), g, ns, out, True, pro)
        else:
            _gather(x, krsc(w), bias, out, g, 1, None, pro)
        return out
    alloc = torch.zeros if cs != Cout else torch.empty
    y = alloc((N, g.DH, g.DW, cs), device=x.device, dtype=_F32)
    hw = g.DH * g.DW
    nimg = N if stats_images is None else int(stats_images)
    if ns > 1:
        _gather_split(x, krsc(w), g, ns, y, False, pro)
        return (y, colsum(y.reshape(-1, Cout)[:nimg * hw], moments=True)) if want_stats else y
    if not want_stats:
        _gather(x, krsc(w), bias, y, g, 0, None, pro)
        return y
    part, G, G1 = _stats_buffer(N * hw, Cout, x.device)
    _gather(x, krsc(w), bias, y, g, 0, part, pro)
    if nimg == N:
        return y, _stats_reduce(part, G, G1, Cout, N * hw)
    # a prefix of the images: their tiles are the first rows of `part` when no tile (128 or 256 pixels) spans two images
    Gs = nimg * hw // CONV_BM
    if hw % 256 == 0 and cs == Cout and (G1 == 0 or Gs % 128 == 0):
        return y, _stats_reduce(part, Gs, 0 if G1 == 0 else Gs // 128, Cout, nimg * hw)
    return y, colsum(y.reshape(-1, Cout)[:nimg * hw], moments=True)


def pack_dgrad_weight(w, koff=0, kw=None):
    """[Cout,R,S,koff:koff+kw] -> [kw,R,S,Cout] for the data-gradient GEMM (whole weight by default)."""
    Cout, Ktot, R, S = w.shape
    kw = kw or Ktot
    o = torch.empty((kw, R, S, Cout), device=w.device, dtype=_F32)
    _call_now("dcs_pack_dgrad_weight", _p(krsc(w)), _p(o), Cout, R, S, kw, Ktot, koff, _stream())
    return o


def conv_dgrad(dy, wp, in_hw, stride, pad, out=None, accumulate=False, src_cs=None, dil=1, bnb=None):
    """Data gradient.  dy [N,OH,OW,cs>=Cout]; wp = pack_dgrad_weight(w) [Cin,R,S,Cout].

    bnb = (y, mask, bn, relu): the result is the gradient arriving at a BatchNorm (input y [N,IH,IW,Cin], record bn,
    ReLU mask from ``mask > 0`` or, with relu, from the recomputed BatchNorm output): the epilogue also reduces the two
    sums of that BatchNorm's backward; returns (out, sums [2,Cin]) -- sums is None when the launch cannot carry them
    (split-K launches of tiny maps), the caller then lets bn_bwd take them in its own pass."""
    _req(dy)
    N = dy.shape[0]
    Cin, R, S, Cout = wp.shape
    IH, IW = in_hw
    gs = geoms_dgrad(N, IH, IW, Cin, Cout, R, S, stride, pad, dil)
    if out is None:
        accumulate = False
        out = (torch.zeros if any(g is None for g in gs) else torch.empty)((N, IH, IW, Cin), device=dy.device, dtype=_F32)
    elif not accumulate and any(g is None for g in gs):
        _defer(out.zero_)
    cs = dy.shape[3]
    fuse = bnb is not None and all(g is not None for g in gs) and out.is_contiguous() and Cin % 4 == 0
    part = None
    if fuse:
        tiles = [-(-(N * g.TY * g.TX) // CONV_BM) for g in gs]
        ns1 = _ksplit(gs[0], N * gs[0].TY * gs[0].TX, Cin) if len(gs) == 1 else 1
        fuse = ns1 == 1
    if fuse:
        G = sum(tiles)
        G1 = 0 if G <= 2048 else -(-G // 128)
        part = (torch.empty((G, 2, Cin), device=dy.device, dtype=_F32) if G1 == 0 else
                torch.zeros((G1 * 128, 2, Cin), device=dy.device, dtype=_F32))
        yb, mb, bnr, relu = bnb
        _req(yb)
        off = 0
        for g, t in zip(gs, tiles):
            if cs != g.src_cstride:
                g = _with_src_cs(g, cs)
            _gather_launch(dy, wp, None, out, g, 1 if accumulate else 0, part[off:], None, (yb, mb, bnr, relu))
            off += t
        sums = torch.empty((2, Cin), device=dy.device, dtype=_F32)
        if G1 == 0:
            _call("dcs_colsum_final", _p(part), _p(sums), 1, G, Cin, 1.0, 0.0, _stream())
        else:
            mid = torch.empty((G1, 2, Cin), device=dy.device, dtype=_F32)
            _call("dcs_colsum_final", _p(part), _p(mid), G1, 128, Cin, 1.0, 0.0, _stream())
            _call("dcs_colsum_final", _p(mid), _p(sums), 1, G1, Cin, 1.0, 0.0, _stream())
        return out, sums
    for g in gs:
        if g is None:
            continue
        if cs != g.src_cstride:
            g = _with_src_cs(g, cs)
        ns = _ksplit(g, N * g.TY * g.TX, Cin) if (len(gs) == 1 and out.is_contiguous()) else 1
        if ns > 1:
            _gather_split(dy, wp, g, ns, out, accumulate)
        else:
            _gather_launch(dy, wp, None, out, g, 1 if accumulate else 0, None)
    return (out, None) if bnb is not None else out


def _with_src_cs(g, cs):
    key = ("cs", id(g), cs)
    h = _geom_cache.get(key)
    if h is None:
        h = DcsConvGeom()
        C.memmove(C.byref(h), C.byref(g), C.sizeof(DcsConvGeom))
        h.src_cstride = cs
        _geom_cache[key] = h
        _geom_cache[("keep", id(g))] = g
    return h


def _nsplit(tiles, M, blocks=1024):
    return max(1, min(-(-blocks // tiles), -(-M // 256)))


def conv_wgrad(x, dy, dw, stride, pad, accumulate, dil=1, koff=None, pro=None):
    """Weight gradient into dw (OIHW channels_last, same layout as the parameter).  koff: x is the input-channel
    slice [koff, koff+Cin) of the convolution, only that slice of dw is written.  pro: as in conv_fwd."""
    _req(x), _req(dy)
    N, H, W, Cin = x.shape
    Cout, Ktot, R, S = dw.shape
    g = geom_fwd(N, H, W, Cin, Cout, R, S, stride, pad, None, None, dil)       # compact slab rows of R*S*Cin
    M = N * g.DH * g.DW
    n = Cout * R * S * Cin
    x3 = Cout % 4 == 0 and os.environ.get("DCS_CONV_X3", "1") != "0" and os.environ.get("DCS_WGRAD_X3", "1") != "0"
    if x3:
        # split-bf16 kernel (csrc/conv_split.hip): one block per (tap, 128x128 or 64x64 tile, split); an EVEN number of
        # splits, whose rounding biases cancel pairwise in the slab reduction
        if R == 3 and S == 3 and stride == 1 and pad == 1 and dil == 1 and W % 16 == 0:
            tiles = (-(-Cout // 64)) * (-(-Cin // 64))      # nine-tap kernel: one block per 64x64 tile and split
            # blocks per launch: 512 (one round of the chip's 2 x 256 block slots) in a step of large maps -- half the slab
            # bytes to write and reduce, for every pyramid level (they share their launch) --, 1024 in a step of small ones
            # (measured: C3 106.9 -> 105.4 ms with 512, C2 17.8-18.7 -> 19.4-19.7; C5 neutral).  "Large": the biggest
            # convolution output of the forward had two million pixels or more.
            nblk = int(os.environ.get("DCS_WGRAD_BLOCKS", "0")) or (512 if _step_max_m[0] >= (2 << 20) else 1024)
            ns = max(2, min(-(-nblk // tiles), (M // 16) // 16))
        else:
            bt = 128 if (Cout > 64 and Cin > 64) else 64
            tiles = R * S * (-(-Cout // bt)) * (-(-Cin // bt))
            ns = _nsplit(tiles, M, 2048 if bt == 64 else 1024)
        ns += ns & 1
        x3 = M >= 64 * ns
    if x3:
        slab = torch.empty((ns, n), device=x.device, dtype=_F32)
        # dy out of bn_bwd carries the device word with its maximum: the rolling 3x3 kernel then runs on two fp16 pieces
        # (the input operand is an activation: only after a training-mode forward, see _x2h_forward)
        dmax = getattr(dy, "_dcs_max", None) if (x2h_on() and _x2h_forward[0]) else None
        _call("dcs_conv_wgrad_x3", _p(x), _p(dy), _p(slab), g, dy.shape[3], 0, ns, _p(pro), _p(dmax), _stream())
    else:
        if R == 3 and S == 3 and stride == 1 and pad == 1 and dil == 1 and W % 32 == 0:
            tiles = (-(-Cout // 64)) * (-(-Cin // 64))          # nine-tap kernel: one block per 64x64 tile and split
            # 1024 blocks = 2 full rounds of 2 blocks per CU, but at least 8 chunks of 32 pixels per block (small inputs)
            ns = max(1, min(-(-1024 // tiles), (M // 32) // 8))
        else:
            bt = 128 if (Cout > 64 and Cin > 64) else 64
            tiles = R * S * (-(-Cout // bt)) * (-(-Cin // bt))
            # 64-wide tiles run 4-5 blocks per CU: twice as many blocks (measured 70.6 -> 81.4 TF on 3x3/2 64->128)
            ns = _nsplit(tiles, M, 2048 if bt == 64 else 1024)
        slab = torch.empty((ns, n), device=x.device, dtype=_F32)
        if pro is None:
            _call("dcs_conv_wgrad", _p(x), _p(dy), _p(slab), C.byref(g), dy.shape[3], 0, ns, _stream())
        else:
            _call("dcs_conv_wgrad_pro", _p(x), _p(dy), _p(slab), C.byref(g), dy.shape[3], 0, ns, _p(pro), _stream())
    if koff is None and Ktot == Cin:
        _call("dcs_reduce_slab", _p(slab), _p(krsc(dw)), n, ns, 1 if accumulate else 0, 0, 0, _stream())
    else:
        dst = C.c_void_p(krsc(dw).data_ptr() + 4 * (koff or 0))
        _call("dcs_reduce_slab", _p(slab), dst, n, ns, 1 if accumulate else 0, Cin, Ktot, _stream())


def pack_stem_weight(w):
    o = torch.empty((w.shape[0], 7, 8, 4), device=w.device, dtype=_F32)
    _call_now("dcs_pack_stem_weight", _p(krsc(w)), _p(o), w.shape[0], 0, _stream())
    return o


def unpack_stem_weight(wp, like, out=None):
    o = out if out is not None else torch.empty_like(like)           # preserves channels_last
    _call("dcs_pack_stem_weight", _p(wp), _p(krsc(o)), wp.shape[0], 1, _stream())
    return o


def stem_conv(p, wp, want_stats=False):
    """7x7/2 pad 3 conv on the NHWC4 normalised image; wp = pack_stem_weight(conv1.weight)."""
    _req(p)
    N, H, W, _ = p.shape
    g = geom_stem_fwd(N, H, W) if os.environ.get("DCS_STEM14", "1") != "0" else geom_stem(N, H, W)
    y = torch.empty((N, g.DH, g.DW, 64), device=p.device, dtype=_F32)
    _step_max_m[0] = max(_step_max_m[0], N * g.DH * g.DW)
    if not want_stats:
        _gather_launch(p, wp, None, y, g, 0, None, fwd=True)
        return y
    part, G, G1 = _stats_buffer(N * g.DH * g.DW, 64, p.device)
    _gather_launch(p, wp, None, y, g, 0, part, fwd=True)
    return y, _stats_reduce(part, G, G1, 64, N * g.DH * g.DW)


def stem_wgrad(p, dy, dwp, accumulate):
    """Accumulates the packed stem weight gradient dwp [64,7,8,4]."""
    N, H, W, _ = p.shape
    g = geom_stem(N, H, W)
    M = N * g.DH * g.DW
    x3 = g.DW % 16 == 0 and M >= 256 and os.environ.get("DCS_CONV_X3", "1") != "0" and os.environ.get("DCS_STEM_X3", "1") != "0"
    if x3:                                     # split-bf16 stem kernel: an even number of splits (bias cancellation)
        ns = max(2, min(512, M // 64))
        ns += ns & 1
        slab = torch.empty((ns, 64 * 224), device=p.device, dtype=_F32)
        dmax = getattr(dy, "_dcs_max", None) if x2h_on() else None     # fp16 two-piece form (the image is bounded)
        _call("dcs_conv_wgrad_x3", _p(p), _p(dy), _p(slab), g, 64, 0, ns, None, _p(dmax), _stream())
        _call("dcs_reduce_slab", _p(slab), _p(dwp), 64 * 224, ns, 1 if accumulate else 0, 0, 0, _stream())
        return
    ns = max(1, min(512, M // 64)) if g.DW % 32 == 0 else _nsplit(7, M)   # seven-row stem kernel: one block per split
    slab = torch.empty((ns, 64 * 224), device=p.device, dtype=_F32)
    _call("dcs_conv_wgrad", _p(p), _p(dy), _p(slab), C.byref(g), 64, 0, ns, _stream())
    _call("dcs_reduce_slab", _p(slab), _p(dwp), 64 * 224, ns, 1 if accumulate else 0, 0, 0, _stream())


def linear(x, w, bias=None):
    """y = x w^T + b as a 1x1 convolution.  x [rows, K]; w [Cout, K] contiguous."""
    rows, K = x.shape
    Cout = w.shape[0]
    g = geom_fwd(rows, 1, 1, K, Cout, 1, 1, 1, 0)
    y = torch.empty((rows, Cout), device=x.device, dtype=_F32)
    _call("dcs_conv_gather", _p(_req(x)), _p(_req(w)), _p(bias), _p(y), C.byref(g), 0, None, _stream())
    return y


def transpose(x):
    R, Cc = x.shape
    o = torch.empty((Cc, R), device=x.device, dtype=_F32)
    _call("dcs_transpose", _p(_req(x)), _p(o), R, Cc, _stream())
    return o


def linear_wgrad(x, dy, dw, accumulate=False):
    """dw [Cout,K] (+)= dy^T x (reduction over the rows, split over row ranges like the conv weight gradient)."""
    rows, K = x.shape
    Cout = dy.shape[1]
    g = geom_fwd(rows, 1, 1, K, Cout, 1, 1, 1, 0)
    bt = 128 if (Cout > 64 and K > 64) else 64
    tiles = (-(-Cout // bt)) * (-(-K // bt))
    ns = max(1, min(-(-512 // tiles), rows // 512))      # about two blocks per CU, >= 16 row chunks per block
    slab = torch.empty((ns, Cout * K), device=x.device, dtype=_F32)
    _call("dcs_conv_wgrad", _p(_req(x)), _p(_req(dy)), _p(slab), C.byref(g), Cout, 0, ns, _stream())
    _call("dcs_reduce_slab", _p(slab), _p(dw), Cout * K, ns, 1 if accumulate else 0, 0, 0, _stream())


# --------------------------------------------------------------------------- #
# reductions / BatchNorm
# --------------------------------------------------------------------------- #
def _groups(rows):
    return int(max(1, min(1024, rows // 64)))


def colsum(x2d, B=1, scale=1.0, moments=False):
    """x2d [B*rows, C(strided)] -> sums [B,2,C] (sum, sum of squares) * scale; moments=True: (mean, biased variance)
    over the rows, formed in double (a ``Moments`` tensor for bn_finalize)."""
    _req(x2d)
    total, Cc = x2d.shape
    rows = total // B
    grp = _groups(rows)
    part = torch.empty((B, grp, 2, Cc), device=x2d.device, dtype=_F32)
    _call("dcs_colsum_partial", _p(x2d), None, None, None, _p(part), B, rows, Cc, Cc, grp, 0, 0, _stream())
    out = torch.empty((B, 2, Cc), device=x2d.device, dtype=_F32)
    _call("dcs_colsum_final", _p(part), _p(out), B, grp, Cc, float(scale), float(rows) if moments else 0.0, _stream())
    return out.as_subclass(Moments) if moments else out


def bn_finalize(sums, gamma, beta, rm, rv, count, training, repeats=1, eps=1e-5, momentum=0.1, update=True):
    Cc = gamma.shape[0]
    bn = torch.empty((4, Cc), device=gamma.device, dtype=_F32)
    upd = training and update
    mode = (2 if isinstance(sums, Moments) else 1) if training else 0
    _call("dcs_bn_finalize", _p(sums), _p(gamma), _p(beta), _p(rm) if (upd or not training) else None,
          _p(rv) if (upd or not training) else None, _p(bn), Cc, float(count), eps, momentum, repeats,
          mode, _stream())
    return bn


def bn_ema_again(bn, rm, rv, count, eps=1e-5, momentum=0.1):
    _call("dcs_bn_ema_again", _p(bn), _p(rm), _p(rv), rm.shape[0], float(count), eps, momentum, _stream())


def bn_act(y, bn, r=None, bn2=None, relu=True):
    _req(y)
    z = torch.empty_like(y)
    Cc = y.shape[-1]
    # a residual block's output in a training forward: its ReLU mask as one byte per float4 beside it -- the backward reads
    # the mask twice (bn_bwd of the block's last BatchNorm, the BatchNorm-backward epilogue of the data gradient that
    # produces its incoming gradient): 1 byte instead of 16 of z each time
    m8 = None
    if relu and r is not None and y.is_cuda and _x2h_forward[0] and mask8_on():
        m8 = torch.empty((y.numel() // 4,), device=y.device, dtype=torch.uint8)
    _call("dcs_bn_act", _p(y), _p(bn), _p(r), _p(bn2), _p(z), y.numel() // Cc, Cc, 1 if relu else 0, _p(m8), _stream())
    if m8 is not None:
        z._mask8 = m8
    return z


def mask8_on():
    """Byte ReLU masks beside residual-block outputs (DCS_MASK8=0: the backward reads the float tensor)."""
    return os.environ.get("DCS_MASK8", "1") != "0"


def bn_bwd(g, y, bn, gamma, masksrc=None, relu=False, want_dy=True, want_gm=False, dy_out=None, acc_dy=False,
           dgamma=None, dbeta=None, acc_param=False, training=True, sums=None):
    """BatchNorm (+ReLU mask) backward.  Returns (dy, gm).  dgamma/dbeta are written (or accumulated).
    sums [2,C] = (sum gm, sum gm * xhat) when the kernel that produced g already reduced them (conv_dgrad(bnb=...))."""
    _req(g), _req(y)
    Cc = y.shape[-1]
    rows = y.numel() // Cc
    if sums is None:
        grp = _groups(rows)
        part = torch.empty((1, grp, 2, Cc), device=y.device, dtype=_F32)
        _call("dcs_colsum_partial", _p(g), _p(y), _p(masksrc), _p(bn), _p(part), 1, rows, Cc, Cc, grp, 1,
              1 if relu else 0, _stream())
        sums = torch.empty((2, Cc), device=y.device, dtype=_F32)
        _call("dcs_colsum_final", _p(part), _p(sums), 1, grp, Cc, 1.0, 0.0, _stream())
    dy = None
    if want_dy:
        dy = dy_out if dy_out is not None else torch.empty_like(y)
    gm = torch.empty_like(y) if want_gm else None
    if want_gm and masksrc is not None and getattr(g, "_dcs_masked", None) is masksrc:
        gm = None                       # g arrived masked (conv_dgrad's epilogue, relu | 4): it IS gm
    smax = None
    if dy is not None and y.is_cuda and x2h_on():
        # the maximum of |dy| rides along (one integer atomicMax per wave): the convolutions that consume dy -- data and
        # weight gradient -- scale it by an exact power of two into fp16 range (csrc/conv_split.hip, two fp16 pieces)
        smax = _max_slot(y.device)
    m8 = getattr(masksrc, "_mask8", None) if (masksrc is not None and mask8_on()) else None
    _call("dcs_bn_bwd_apply", _p(g), _p(y), None if m8 is not None else _p(masksrc), _p(bn), _p(gamma), _p(sums), _p(dy),
          _p(gm), _p(dgamma), _p(dbeta), rows, Cc, 1 if relu else 0, 1 if (acc_dy and dy_out is not None) else 0, 0,
          1 if acc_param else 0, 1 if training else 0, _p(smax), _p(m8), _stream())
    if smax is not None:
        dy._dcs_max = smax
    if want_gm and gm is None:
        gm = g
    return dy, gm


# --------------------------------------------------------------------------- #
# pyramid / pooling / resize
# --------------------------------------------------------------------------- #
def normalize_pyramid(img, mean3, std3, levels=3):
    """levels=1: only the full-resolution NHWC4 image (DeepLab: no pyramid; mean 0 / std 1 = plain repack).
    img: NCHW fp32 device tensor, or a list of such tensors (batch parts, e.g. the two crops): the parts are
    normalised straight into one NHWC4 batch, so the caller needs no torch.cat copy."""
    parts = list(img) if isinstance(img, (list, tuple)) else [img]
    for t in parts:
        if not t.is_cuda or t.dtype != _F32:
            raise RuntimeError("normalize_pyramid needs fp32 device tensors (no CPU fallback)")
    parts = [t.contiguous() for t in parts]
    N = sum(t.shape[0] for t in parts)
    _, Cc, H, W = parts[0].shape
    assert Cc == 3 and all(t.shape[1:] == parts[0].shape[1:] for t in parts)
    mk = lambda h, w: torch.empty((N, h, w, 4), device=parts[0].device, dtype=_F32)
    p0 = mk(H, W)
    p1, p2 = (mk(H // 2, W // 2), mk(H // 4, W // 4)) if levels == 3 else (None, None)
    b = 0
    for t in parts:
        n = t.shape[0]
        _call("dcs_normalize_pyramid", _p(t), _p(p0[b:b + n]), _p(p1[b:b + n]) if p1 is not None else None,
              _p(p2[b:b + n]) if p2 is not None else None, n, H, W, _p(mean3), _p(std3), _stream())
        b += n
    return p0, p1, p2


def bn_relu_maxpool(y, bn):
    N, H, W, Cc = y.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, OH, OW, Cc), device=y.device, dtype=_F32)
    idx = torch.empty((N, OH, OW, Cc), device=y.device, dtype=torch.uint8)
    _call("dcs_bn_relu_maxpool", _p(_req(y)), _p(bn), _p(out), _p(idx), N, H, W, Cc, _stream())
    idx._pooled = out                    # bn_pool_bwd takes its sums from the pooled tensors
    return out, idx


def maxpool_bwd(g, idx, H, W):
    N, _, _, Cc = g.shape
    gz = torch.empty((N, H, W, Cc), device=g.device, dtype=_F32)
    _call("dcs_maxpool_bwd", _p(_req(g)), _p(idx), _p(gz), N, H, W, Cc, _stream())
    return gz


def bn_pool_bwd(g, idx, y, bn, gamma, dgamma=None, dbeta=None, acc_param=False, training=True):
    """Backward of bn -> relu -> maxpool3x3/2 in one step: g [N,OH,OW,C] is the gradient of the POOLED map, idx the
    saved argmax; returns dy [N,H,W,C] (gradient of the conv output y).  The dense gradient of the un-pooled map is
    never written (it is re-gathered from g/idx in both BatchNorm passes)."""
    _req(g), _req(y)
    N, H, W, Cc = y.shape
    grp = _groups((N * ((H + 1) // 2) * ((W + 1) // 2)) // 2)
    part = torch.empty((grp, 2, Cc), device=y.device, dtype=_F32)
    z = getattr(idx, "_pooled", None)
    if z is not None and os.environ.get("DCS_POOL_SUMS", "1") != "0":
        # the pooled output is beside the argmax indices: the two sums need only the pooled tensors (1/4 of the map)
        _call("dcs_bn_pool_bwd_partial_pooled", _p(g), _p(z), _p(idx), _p(y), _p(bn), _p(part), N, H, W, Cc, grp, _stream())
    else:
        _call("dcs_bn_pool_bwd_partial", _p(g), _p(idx), _p(y), _p(bn), _p(part), N, H, W, Cc, grp, _stream())
    sums = torch.empty((2, Cc), device=y.device, dtype=_F32)
    _call("dcs_colsum_final", _p(part), _p(sums), 1, grp, Cc, 1.0, 0.0, _stream())
    dy = torch.empty_like(y)
    smax = _max_slot(y.device) if (y.is_cuda and x2h_on()) else None     # max |dy|: fp16 two-piece stem weight gradient
    _call("dcs_bn_pool_bwd_apply", _p(g), _p(idx), _p(y), _p(bn), _p(gamma), _p(sums), _p(dy), _p(dgamma), _p(dbeta),
          N, H, W, Cc, 1 if acc_param else 0, 1 if training else 0, _p(smax), _stream())
    if smax is not None:
        dy._dcs_max = smax
    return dy


def upsample_add(x, skips: Sequence[torch.Tensor], OH, OW, want_stats=False):
    """t = bilinear_upsample(x, (OH, OW)) + sum(skips).  want_stats: also the batch statistics of t (Moments, as from
    conv_fwd(want_stats=True)) reduced by the same kernel instead of a colsum pass over t."""
    N, IH, IW, Cc = x.shape
    t = torch.empty((N, OH, OW, Cc), device=x.device, dtype=_F32)
    s = list(skips) + [None] * (3 - len(skips))
    rows = N * OH * OW
    if want_stats and Cc <= 1024 and 256 % (Cc // 4) == 0 and os.environ.get("DCS_UPSAMPLE_STATS", "1") != "0":
        groups = int(max(1, min(2048, (rows * (Cc // 4)) // 256)))
        part = torch.empty((groups, 2, Cc), device=x.device, dtype=_F32)
        _call("dcs_upsample_add_stats", _p(_req(x)), _p(s[0]), _p(s[1]), _p(s[2]), _p(t), _p(part), groups, N, IH, IW, OH, OW,
              Cc, _stream())
        return t, _stats_reduce(part, groups, 0, Cc, rows)
    _call("dcs_upsample_add", _p(_req(x)), _p(s[0]), _p(s[1]), _p(s[2]), _p(t), N, IH, IW, OH, OW, Cc, _stream())
    return (t, colsum(t.reshape(-1, Cc), moments=True)) if want_stats else t


def upsample_bwd(g, IH, IW, out=None, accumulate=False):
    N, OH, OW, Cc = g.shape
    if out is None:
        out = torch.empty((N, IH, IW, Cc), device=g.device, dtype=_F32)
        accumulate = False
    smax = _max_slot(g.device) if (g.is_cuda and x2h_on() and not accumulate) else None    # max |out| rides along (tag_max)
    _call("dcs_upsample_bwd", _p(_req(g)), _p(out), N, IH, IW, OH, OW, Cc, 1 if accumulate else 0, _p(smax), _stream())
    if smax is not None:
        out._dcs_max = smax
    return out


def upsample_to_nchw(x, Cc, OH, OW):
    N, IH, IW, cs = x.shape
    out = torch.empty((N, Cc, OH, OW), device=x.device, dtype=_F32)
    _call("dcs_upsample_to_nchw", _p(_req(x)), _p(out), N, IH, IW, cs, Cc, OH, OW, _stream())
    return out


def upsample_to_nchw_bwd(g, IH, IW, cs, gscale=None):
    N, Cc, OH, OW = g.shape
    gx = torch.empty((N, IH, IW, cs), device=g.device, dtype=_F32)
    tmp = torch.empty((N, Cc, OH, IW), device=g.device, dtype=_F32) if Cc <= 32 else None
    _call("dcs_upsample_to_nchw_bwd", _p(_req(g)), _p(gscale), _p(gx), _p(tmp), N, IH, IW, cs, Cc, OH, OW, _stream())
    return gx


# --------------------------------------------------------------------------- #
# losses
# --------------------------------------------------------------------------- #
SEG_MODES = {"full": 0, "plain_focal": 1, "no_class_weights": 2, "no_EDT": 3, "ce": 4}


def seg_loss(logits, target, ldw, cw, mode, gamma=0.5, ignore=255):
    """Returns (out[3] = loss, count, 1/count ; grad = unscaled d(sum loss)/d logits)."""
    _req(logits)
    if target.dtype != torch.int64 or not target.is_contiguous() or not target.is_cuda:
        raise RuntimeError("seg_loss: target must be a contiguous int64 device tensor")
    N, Cc, H, W = logits.shape
    grad = torch.empty_like(logits)
    blocks = int(max(1, min(4096, (N * H * W + 255) // 256)))
    part = torch.empty((blocks, 2), device=logits.device, dtype=_F32)
    out = torch.empty((3,), device=logits.device, dtype=_F32)
    _call("dcs_seg_loss", _p(logits), _p(target), _p(ldw), _p(cw), _p(grad), _p(part), N, Cc, H, W,
          SEG_MODES[mode], float(gamma), int(ignore), blocks, _stream())
    _call("dcs_seg_loss_final", _p(part), _p(out), blocks, _stream())
    return out, grad


def seg_loss_fused_ok(ih, iw, H, W, Cc):
    """The fused upsample + loss kernel covers exact x2 / x4 upsampling of <= 20 classes."""
    return Cc <= 20 and H % ih == 0 and W % iw == 0 and H // ih == W // iw and H // ih in (2, 4)


def seg_loss_fused(logits_lr, Cc, target, ldw, cw, mode, gamma=0.5, ignore=255):
    """logits_lr: NHWC low-resolution logits [N,ih,iw,cs] (first Cc channels used); target int64 [N,H,W].
    Returns (out[3] = loss, count, 1/count ; grad_lr [N,ih,iw,cs] = unscaled d(sum loss)/d logits_lr)."""
    _req(logits_lr)
    if target.dtype != torch.int64 or not target.is_contiguous() or not target.is_cuda:
        raise RuntimeError("seg_loss: target must be a contiguous int64 device tensor")
    N, ih, iw, cs = logits_lr.shape
    H, W = target.shape[1:]
    grad = torch.empty_like(logits_lr)
    blocks = _lib.load().dcs_seg_loss_fused_blocks(N, ih, iw)
    if blocks <= 0:
        _lib.check(blocks, "dcs_seg_loss_fused_blocks")
    part = torch.empty((blocks, 2), device=logits_lr.device, dtype=_F32)
    out = torch.empty((3,), device=logits_lr.device, dtype=_F32)
    _call("dcs_seg_loss_fused", _p(logits_lr), cs, _p(target), _p(ldw), _p(cw), _p(grad), _p(part), N, Cc, ih, iw, H, W,
          SEG_MODES[mode], float(gamma), int(ignore), blocks, _stream())
    _call("dcs_seg_loss_final", _p(part), _p(out), blocks, _stream())
    return out, grad


def scale_inplace(x, a, b=None):
    _call("dcs_scale_inplace", _p(x), x.numel(), _p(a), _p(b), _stream())
    return x


ANCHOR_CHUNK = 1024


def anchor_keys_raw(logits, N, h, w, cs, Cc, labels, ignore=255):
    """logits: fp32 device tensor whose storage from data_ptr() is [N,h,w,cs] (first Cc channels used);
    labels int64 [N,H,W] -> (key uint8 [N,h*w], hist int32 [N,nchunks,2C])."""
    if not logits.is_cuda or logits.dtype != _F32 or labels.dtype != torch.int64 or not labels.is_contiguous():
        raise RuntimeError("anchor_keys: fp32 device logits and contiguous int64 labels required")
    H, W = labels.shape[1:]
    nch = -(-(h * w) // ANCHOR_CHUNK)
    key = torch.empty((N, h * w), device=labels.device, dtype=torch.uint8)
    hist = torch.empty((N, nch, 2 * Cc), device=labels.device, dtype=torch.int32)
    _call("dcs_anchor_keys", _p(logits), cs, Cc, _p(labels), N, h, w, H, W, ignore, _p(key), _p(hist),
          ANCHOR_CHUNK, _stream())
    return key, hist


def anchor_select(key, hist, req, Cc):
    """req int32 [Q,3] = (image, key, rank) -> flat pixel index int32 [Q]."""
    N, HW = key.shape
    Q = req.shape[0]
    out = torch.empty((Q,), device=key.device, dtype=torch.int32)
    _call("dcs_anchor_select", _p(key), _p(hist), _p(req), _p(out), Q, N, HW, Cc, ANCHOR_CHUNK, _stream())
    return out


def gather_rows(feat2d, rowidx):
    A = rowidx.shape[0]
    Cc = feat2d.shape[-1]
    X = torch.empty((A, Cc), device=feat2d.device, dtype=_F32)
    _call("dcs_gather_rows", _p(feat2d), _p(rowidx), _p(X), A, Cc, _stream())
    return X


def scatter_add_rows(gX, rowidx, gfeat):
    _call("dcs_scatter_add_rows", _p(_req(gX)), _p(rowidx), _p(gfeat), rowidx.shape[0], gX.shape[1], _stream())


def gather_rows_bilinear(feat, rowidx, OH, OW):
    """feat NHWC [N,IH,IW,C]; rowidx int32 [A] into the virtual upsampled [N,OH,OW] grid -> X [A,C]."""
    _req(feat)
    N, IH, IW, Cc = feat.shape
    A = rowidx.shape[0]
    X = torch.empty((A, Cc), device=feat.device, dtype=_F32)
    _call("dcs_gather_rows_bilinear", _p(feat), _p(rowidx), _p(X), A, Cc, N, IH, IW, OH, OW, _stream())
    return X


def scatter_rows_bilinear(gX, rowidx, gfeat, OH, OW):
    """gfeat NHWC [N,IH,IW,C] += adjoint of gather_rows_bilinear."""
    N, IH, IW, Cc = gfeat.shape
    _call("dcs_scatter_rows_bilinear", _p(_req(gX)), _p(rowidx), _p(_req(gfeat)), rowidx.shape[0], Cc, N, IH, IW, OH, OW,
          _stream())


_contrast_sync = {}          # device index -> two zeroed uint32 (the grid barrier of the one-launch small family)


def contrast_fwd_bwd(X, labels, mode, temperature=0.07, mask=None):
    """Contrastive loss on anchors X [A,C] (view-major) with float labels [A]; forward and backward in one library call
    (dcs_contrast_fused: one launch for A <= 1024).

    mode 0: pixel contrast (utils/loss.py:339-389); 1: SupCon/SimCLR (:175-204).  X may be a row-strided view
    (X.stride(1) == 1, X.stride(0) % 4 == 0) and labels a strided 1-D view -- e.g. columns of the packed all-gather
    buffer of the data-parallel step; rows with label < 0 are padding.  mask: optional explicit [b,b] positive weights
    (SupConLoss(mask=...), mode 1).  Returns (loss [1], dX [A,C] = d loss / d X)."""
    if not (X.is_cuda and X.dtype == _F32 and X.dim() == 2 and X.stride(1) == 1 and X.stride(0) % 4 == 0):
        raise RuntimeError("contrast_fwd_bwd needs a fp32 device matrix with unit column stride (no CPU fallback)")
    if not (labels.is_cuda and labels.dtype == _F32 and labels.dim() == 1):
        raise RuntimeError("contrast_fwd_bwd: labels must be a 1-D fp32 device tensor")
    A, Cc = X.shape
    ldx, ldy = X.stride(0), labels.stride(0) if A > 1 else 1
    mb = 0
    if mask is not None:
        mask = _req(mask.to(_F32).contiguous())
        mb = mask.shape[0]
        assert mode == 1 and mask.shape == (mb, mb) and A % mb == 0
    need = C.c_int64(0)
    _call("dcs_contrast_fused_ws", A, Cc, C.byref(need))
    ws = torch.empty((need.value,), device=X.device, dtype=_F32)
    loss = torch.empty((1,), device=X.device, dtype=_F32)
    it = 1.0 / temperature
    sync = _contrast_sync.get(X.device.index)
    if sync is None:
        sync = _contrast_sync[X.device.index] = torch.zeros((2,), device=X.device, dtype=torch.int32)
    if Cc <= 128:
        dX = torch.empty((A, Cc), device=X.device, dtype=_F32)
        _call("dcs_contrast_fused", _p(X), ldx, _p(labels), ldy, _p(mask), mb, A, Cc, mode, it, _p(loss), _p(dX), Cc, None, 0,
              _p(ws), need.value, _p(sync), _stream())
        return loss, dX
    # wide features (DeepLab's 2048-channel pixel contrast): the kernel hands back Gs = G + G^T [A, ld] and dX = Gs X is
    # one GEMM with K = ld (zero padded), weights X^T
    ld = -(-A // 32) * 32
    Gs = torch.zeros((A, ld), device=X.device, dtype=_F32) if ld != A else torch.empty((A, ld), device=X.device, dtype=_F32)
    _call("dcs_contrast_fused", _p(X), ldx, _p(labels), ldy, _p(mask), mb, A, Cc, mode, it, _p(loss), None, 0, _p(Gs), ld,
          _p(ws), need.value, _p(sync), _stream())
    Xp = torch.zeros((ld, Cc), device=X.device, dtype=_F32)
    Xp[:A].copy_(X)
    Xt = transpose(Xp)                                   # [C, ld]
    g2 = geom_fwd(A, 1, 1, ld, Cc, 1, 1, 1, 0)
    dX = torch.empty((A, Cc), device=X.device, dtype=_F32)
    _call("dcs_conv_gather", _p(Gs), _p(Xt), None, _p(dX), C.byref(g2), 0, None, _stream())
    return loss, dX


def contrast_fwd_bwd_unfused(X, labels, mode, temperature=0.07):
    """The round-1 form (GEMM writes S, row kernel, symmetrize, GEMM): kept only as the A/B baseline of bench.py."""
    _req(X)
    A, Cc = X.shape
    ld = -(-A // 32) * 32
    g = geom_fwd(A, 1, 1, Cc, A, 1, 1, 1, 0, None, ld)
    S = torch.empty((A, ld), device=X.device, dtype=_F32)
    _call("dcs_conv_gather", _p(X), _p(X), None, _p(S), C.byref(g), 0, None, _stream())
    loss_row = torch.empty((A,), device=X.device, dtype=_F32)
    G = torch.empty((A, ld), device=X.device, dtype=_F32)
    _call("dcs_contrast_rows", _p(S), _p(labels), _p(loss_row), _p(G), A, ld, mode, 1.0 / temperature, _stream())
    loss = torch.empty((1,), device=X.device, dtype=_F32)
    _call("dcs_sum_scalar", _p(loss_row), _p(loss), A, 1.0 / A, _stream())
    Gs = torch.empty((A, ld), device=X.device, dtype=_F32)
    _call("dcs_symmetrize", _p(G), _p(Gs), A, ld, _stream())
    if A > 1024:
        dXp = torch.empty((ld, Cc), device=X.device, dtype=_F32)
        linear_wgrad(X, Gs, dXp)
        return loss, dXp[:A]
    Xp = X
    if ld != A:
        Xp = torch.zeros((ld, Cc), device=X.device, dtype=_F32)
        Xp[:A].copy_(X)
    Xt = transpose(Xp)                                   # [C, ld]
    g2 = geom_fwd(A, 1, 1, ld, Cc, 1, 1, 1, 0)
    dX = torch.empty((A, Cc), device=X.device, dtype=_F32)
    _call("dcs_conv_gather", _p(Gs), _p(Xt), None, _p(dX), C.byref(g2), 0, None, _stream())
    return loss, dX


def dropout(x, p, noise=None, seed=0):
    """nn.Dropout forward.  noise: optional float 0/1 keep tensor (same shape, e.g. drawn on the host from the CPU
    generator like the reference); otherwise a device-side counter-based mask from ``seed``.  -> (out, mask uint8)."""
    _req(x)
    out = torch.empty_like(x)
    mask = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    _call("dcs_dropout", _p(x), _p(noise), _p(mask), _p(out), x.numel(), float(p), int(seed) & 0xFFFFFFFF, _stream())
    return out, mask


def dropout_bwd(g, mask, p):
    out = torch.empty_like(g)
    _call("dcs_dropout_bwd", _p(_req(g)), _p(mask), _p(out), g.numel(), float(p), _stream())
    return out


def confusion(logits, labels, num_classes, conf, lowres=None, want_pred=False):
    """conf [N,C,C] int64 (device) += confusion of argmax(logits) vs labels.  logits: NCHW full-resolution tensor, or,
    with lowres=(H, W), the low-resolution NHWC logits buffer that is upsampled on the fly to HxW."""
    N = labels.shape[0]
    H, W = labels.shape[1:]
    if labels.dtype != torch.int64 or not labels.is_contiguous() or not labels.is_cuda:
        raise RuntimeError("confusion: labels must be a contiguous int64 device tensor (no CPU fallback)")
    pred = torch.empty((N, H, W), device=labels.device, dtype=torch.uint8) if want_pred else None
    if lowres is None:
        _req(logits)
        _call("dcs_confusion", _p(logits), _p(labels), _p(pred), _p(conf), N, num_classes, H, W, 0, 0, 0, _stream())
    else:
        _req(logits)
        _, ih, iw, cs = logits.shape
        _call("dcs_confusion", _p(logits), _p(labels), _p(pred), _p(conf), N, num_classes, H, W, ih, iw, cs, _stream())
    return pred


def label_boundary_weights(labels, num_classes, ignore_id=255):
    """labels int64 [B,H,W] on the device -> (weight fp32 [B,H,W], dist int32 [B,H,W] in 16.16 fixed point)."""
    if labels.dtype != torch.int64 or not labels.is_contiguous() or not labels.is_cuda or labels.dim() != 3:
        raise RuntimeError("label_boundary_weights: labels must be a contiguous int64 [B,H,W] device tensor "
                           "(no CPU fallback)")
    B, H, W = labels.shape
    dist = torch.empty((B, H, W), device=labels.device, dtype=torch.int32)
    weight = torch.empty((B, H, W), device=labels.device, dtype=_F32)
    img_std = torch.empty((B,), device=labels.device, dtype=_F32)
    _call("dcs_label_boundary_weights", _p(labels), _p(dist), _p(img_std), _p(weight), B, H, W, int(num_classes),
          int(ignore_id), _stream())
    return weight, dist


def sum_scalar(x, scale=1.0):
    out = torch.empty((1,), device=x.device, dtype=_F32)
    _call("dcs_sum_scalar", _p(_req(x)), _p(out), x.numel(), float(scale), _stream())
    return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, wd, step):
    _call("dcs_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, wd, step, _stream())


def axpy(y, x, a):
    _call("dcs_axpy", _p(y), _p(x), x.numel(), float(a), _stream())


def add_rowvec_bcast(g, v, scale, accumulate=True):
    N, H, W, Cc = g.shape
    _call("dcs_add_rowvec_bcast", _p(g), _p(_req(v)), N, H * W, Cc, float(scale), 1 if accumulate else 0, _stream())


def relu_bwd(g, z):
    out = torch.empty_like(g)
    _call("dcs_relu_bwd_rows", _p(_req(g)), _p(_req(z)), _p(out), g.numel(), _stream())
    return out
