"""CPU: the launch recorder behind ops.level_batch (dcs_amd/ops.py).  No kernel runs: the library entry points are replaced
by a logger, and the properties the model relies on are checked on the emitted launch sequence --
  * every level's launches keep their own order,
  * launches of different levels that name the same shared memory (running statistics, shared gradients) keep the order
    in which they were recorded,
  * convolution launches standing at the head of several levels leave through ONE *_multi call carrying all of them,
  * levels whose sequences differ (another kernel family on a small map) still drain completely."""
import ctypes as C
import random

import pytest

import dcs_amd.ops as ops
from dcs_amd.lib import DcsConvGeom


@pytest.fixture()
def log(monkeypatch):
    calls = []
    monkeypatch.setattr(ops, "_call_now", lambda name, *a: calls.append((name, a)))
    monkeypatch.setenv("DCS_LEVEL_BATCH", "1")
    return calls


def _gather(tag):
    g = DcsConvGeom()
    g.N = tag
    return ("dcs_conv_gather_x3", C.c_void_p(tag), C.c_void_p(1), None, C.c_void_p(2), g, 0, None, None, None, None, None, 0, 1, 0,
            None, C.c_void_p(0))


def _wgrad(tag):
    g = DcsConvGeom()
    g.N = tag
    return ("dcs_conv_wgrad_x3", C.c_void_p(tag), C.c_void_p(1), C.c_void_p(2), g, 64, 0, 2, None, None, C.c_void_p(0))


def _flat(calls):
    """emitted launches as (entry, tag) with multi launches expanded in place"""
    out = []
    for name, a in calls:
        if name.endswith("_multi"):
            out += [(name[:-6], a[0][i].src) for i in range(a[1])]
        elif name in ("dcs_conv_gather_x3", "dcs_conv_wgrad_x3"):
            out.append((name, a[0].value))
        else:
            out.append((name, a[0]))
    return out


def test_three_identical_levels_merge_every_convolution(log):
    with ops.level_batch() as lb:
        for lvl in range(3):
            lb.level(lvl)
            ops._call(*_gather(100 + lvl))
            ops._call("dcs_colsum_final", 200 + lvl)
            ops._call("dcs_bn_finalize", 300 + lvl, None, None, C.c_void_p(0xAA), None)       # shared running statistics
            ops._call(*_gather(400 + lvl))
        lb.flush()
        assert [n for n, _ in log] == ["dcs_conv_gather_x3_multi", "dcs_colsum_final", "dcs_bn_finalize", "dcs_colsum_final",
                                       "dcs_bn_finalize", "dcs_colsum_final", "dcs_bn_finalize", "dcs_conv_gather_x3_multi"]
        assert [a[1] for n, a in log if n.endswith("_multi")] == [3, 3]
        assert [a[0] for n, a in log if n == "dcs_bn_finalize"] == [300, 301, 302]           # recorded order
        lb.level(0)
        ops._call("dcs_colsum_final", 1)
    assert log[-1] == ("dcs_colsum_final", (1,))                                             # leaving the block flushes


def test_shared_memory_updates_keep_their_recorded_order_when_levels_run_in_reverse(log):
    """The backward visits the levels 2, 1, 0: level 2's BatchNorm-parameter gradient and weight-gradient reduction come
    first (overwrite), the others accumulate -- and the three weight-gradient launches still leave as one."""
    with ops.level_batch() as lb:
        for lvl in (2, 1, 0):
            lb.level(lvl)
            ops._call("dcs_bn_bwd_apply", 10 + lvl, None, None, None, None, None, None, None, C.c_void_p(0xBB), None)
            ops._call(*_wgrad(20 + lvl))
            ops._call("dcs_reduce_slab", 30 + lvl, C.c_void_p(0xCC))
        lb.flush()
    names = [n for n, _ in log]
    assert names.count("dcs_conv_wgrad_x3_multi") == 1 and "dcs_conv_wgrad_x3" not in names
    assert [a[0] for n, a in log if n == "dcs_bn_bwd_apply"] == [12, 11, 10]
    assert [a[0] for n, a in log if n == "dcs_reduce_slab"] == [32, 31, 30]
    assert names.index("dcs_conv_wgrad_x3_multi") > max(i for i, n in enumerate(names) if n == "dcs_bn_bwd_apply")


def test_levels_on_different_kernel_families_drain_and_merge_what_they_share(log):
    with ops.level_batch() as lb:
        lb.level(0)
        ops._call("dcs_conv3x3_x3w", *_gather(1)[1:13], C.c_void_p(0))          # large map: halo kernel
        ops._call("dcs_colsum_final", 5)
        for lvl in (1, 2):
            lb.level(lvl)
            ops._call(*_gather(1 + lvl))                                           # small maps: per-tap kernel
            ops._call("dcs_colsum_final", 5 + lvl)
        lb.flush()
    names = [n for n, _ in log]
    assert names.count("dcs_conv_gather_x3_multi") == 1 and names.count("dcs_conv3x3_x3w") == 1
    assert sorted(a[0] for n, a in log if n == "dcs_colsum_final") == [5, 6, 7]


@pytest.mark.parametrize("seed", range(8))
def test_random_sequences_keep_level_order_and_key_order(log, seed):
    rnd = random.Random(seed)
    recorded = []
    with ops.level_batch() as lb:
        for step in range(4):
            for lvl in rnd.sample(range(3), 3):
                lb.level(lvl)
                for _ in range(rnd.randint(1, 6)):
                    tag = len(recorded) + 1000
                    kind = rnd.choice(["gather", "wgrad", "plain", "keyed", "keyed"])
                    if kind == "gather":
                        ops._call(*_gather(tag))
                        recorded.append((lvl, "dcs_conv_gather_x3", tag, None))
                    elif kind == "wgrad":
                        ops._call(*_wgrad(tag))
                        recorded.append((lvl, "dcs_conv_wgrad_x3", tag, None))
                    elif kind == "plain":
                        ops._call("dcs_bn_act", tag)
                        recorded.append((lvl, "dcs_bn_act", tag, None))
                    else:
                        key = rnd.choice([0xA0, 0xB0])
                        ops._call("dcs_reduce_slab", tag, C.c_void_p(key))
                        recorded.append((lvl, "dcs_reduce_slab", tag, key))
            lb.flush()
    flat = _flat(log)
    assert sorted(flat) == sorted((n, t) for _, n, t, _ in recorded)                 # everything left, once
    pos = {t: i for i, (_, t) in enumerate(flat)}
    for lvl in range(3):                                                             # per-level order
        tags = [t for l, _, t, _ in recorded if l == lvl]
        assert [pos[t] for t in tags] == sorted(pos[t] for t in tags)
    for key in (0xA0, 0xB0):                                                         # shared-memory order
        tags = [t for _, _, t, k in recorded if k == key]
        assert [pos[t] for t in tags] == sorted(pos[t] for t in tags)


def test_switch_off_and_nesting(log, monkeypatch):
    monkeypatch.setenv("DCS_LEVEL_BATCH", "0")
    with ops.level_batch() as lb:
        lb.level(1)
        ops._call("dcs_bn_act", 1)
        assert log == [("dcs_bn_act", (1,))]                                         # at once
        lb.flush()
    monkeypatch.setenv("DCS_LEVEL_BATCH", "1")
    with ops.level_batch() as outer:
        with ops.level_batch() as inner:                                             # nested: the outer one records
            inner.level(2)
            ops._call("dcs_bn_act", 2)
            inner.flush()
        assert len(log) == 1
        outer.flush()
    assert log[-1] == ("dcs_bn_act", (2,))
    assert ops._batch is None
