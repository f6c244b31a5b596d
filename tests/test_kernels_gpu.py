"""GPU: every kernel of libdcs_hip.so, called through the C ABI (dcs_amd.ops -> ctypes), against the torch-CPU
statement of its contract (tests/emu_ops.py) on seeded inputs, including odd extents, ragged tiles
(M, Cout, K not multiples of the tile sizes), stride-2 parity classes and accumulate modes.
Tolerances: fp32 kernels vs fp32/fp64 CPU math -> 2e-5 relative to the tensor's max (written per test)."""
import numpy as np
import pytest
import torch

import emu_ops as E

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import dcs_amd.ops as real
    real._lib.load()
    return real


def rnd(*shape, seed=0, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def cl(w):
    return w.contiguous(memory_format=torch.channels_last)


def close(a, b, tol=2e-5, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-20)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, (what, err)


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride
    (2, 9, 13, 64, 64, 3, 1),
    (1, 16, 24, 64, 128, 3, 2),
    (2, 7, 11, 128, 128, 3, 1),
    (1, 5, 6, 256, 512, 3, 2),
    (2, 8, 8, 64, 128, 1, 1),
    (1, 9, 7, 128, 256, 1, 2),
    (3, 6, 10, 512, 128, 1, 1),
    (1, 33, 17, 128, 128, 3, 1),
    (1, 1, 2, 512, 512, 3, 1),
    (2, 6, 64, 64, 64, 3, 1),        # W % 32 == 0: nine-tap weight-gradient kernel
    (1, 5, 32, 128, 64, 3, 1),
    (3, 3, 96, 64, 128, 3, 1),
    (1, 2, 32, 256, 256, 3, 1),
    (3, 300, 320, 64, 64, 1, 1),     # 2250 row tiles: two-level reduction of the fused BN statistics
    (2, 8, 64, 64, 64, 3, 1),
    (1, 12, 32, 32, 40, 3, 1),       # ragged output channels
    (2, 4, 96, 16, 64, 3, 1),        # a single 16-channel K chunk per tap
    (1, 8, 32, 128, 128, 3, 1),
    (8, 16, 32, 256, 256, 3, 1),     # 4096 pixels x 2 column tiles x 72 chunks: split-K forward and data gradient
    (4, 24, 24, 512, 512, 3, 1),     # 2304 pixels, ragged last row tile
    (8, 16, 32, 128, 256, 3, 2),     # strided: forward splits, the four parity-class data gradients do not
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(ops, N, H, W, Cin, Cout, k, s):
    x = rnd(N, H, W, Cin, seed=1)
    w = cl(rnd(Cout, Cin, k, k, seed=2, scale=0.05))
    pad = k // 2
    y_ref = E.conv_fwd(x, w, s, pad)
    y = ops.conv_fwd(x.to(DEV), cl(w.to(DEV)), s, pad)
    close(y, y_ref, what="fwd")
    y2, sums = ops.conv_fwd(x.to(DEV), cl(w.to(DEV)), s, pad, want_stats=True)
    close(y2, y_ref, what="fwd with stats epilogue")
    from dcs_amd.ops import Moments
    assert isinstance(sums, Moments)        # (mean, biased variance), formed in double from the epilogue's tile sums
    close(sums[0, 0], E.colsum(y_ref.reshape(-1, Cout), moments=True)[0, 0], 1e-5, "fused BN statistics: mean")
    close(sums[0, 1], E.colsum(y_ref.reshape(-1, Cout), moments=True)[0, 1], 1e-5, "fused BN statistics: variance")
    dy = rnd(*y_ref.shape, seed=3)
    wp_ref = E.pack_dgrad_weight(w)
    wp = ops.pack_dgrad_weight(cl(w.to(DEV)))
    close(wp, wp_ref, 0.0, "pack")
    gx_ref = E.conv_dgrad(dy, wp_ref, (H, W), s, pad)
    gx = ops.conv_dgrad(dy.to(DEV), wp, (H, W), s, pad)
    close(gx, gx_ref, what="dgrad")
    base = rnd(N, H, W, Cin, seed=4)
    acc = base.to(DEV).clone()
    ops.conv_dgrad(dy.to(DEV), wp, (H, W), s, pad, out=acc, accumulate=True)
    close(acc, base + gx_ref, what="dgrad accumulate")
    dw_ref = torch.empty_like(w)
    E.conv_wgrad(x, dy, dw_ref, s, pad, False)
    dw = torch.empty_like(cl(w.to(DEV)))
    ops.conv_wgrad(x.to(DEV), dy.to(DEV), dw, s, pad, False)
    close(dw, dw_ref, 5e-5, "wgrad")
    ops.conv_wgrad(x.to(DEV), dy.to(DEV), dw, s, pad, True)
    close(dw, 2 * dw_ref, 5e-5, "wgrad accumulate")


def test_conv_seg_head_19_classes_bias_and_padded_stride(ops):
    x = rnd(2, 10, 12, 128, seed=5)
    w = cl(rnd(19, 128, 1, 1, seed=6, scale=0.1))
    b = rnd(19, seed=7)
    y = ops.conv_fwd(x.to(DEV), cl(w.to(DEV)), 1, 0, bias=b.to(DEV), dst_cs=20)
    ref = E.conv_fwd(x, w, 1, 0, bias=b, dst_cs=20)
    close(y, ref, what="fwd cs20")
    assert float(y[..., 19].abs().max()) == 0.0
    dy = torch.zeros(2, 10, 12, 20)
    dy[..., :19] = rnd(2, 10, 12, 19, seed=8)
    dw = torch.empty_like(cl(w.to(DEV)))
    ops.conv_wgrad(x.to(DEV), dy.to(DEV), dw, 1, 0, False)
    dref = torch.empty_like(w)
    E.conv_wgrad(x, dy, dref, 1, 0, False)
    close(dw, dref, 5e-5, "wgrad cs20")
    wpad = torch.zeros(128, 1, 1, 20)
    wpad[..., :19] = E.pack_dgrad_weight(w)
    gx = ops.conv_dgrad(dy.to(DEV), wpad.to(DEV), (10, 12), 1, 0)
    close(gx, E.conv_dgrad(dy, E.pack_dgrad_weight(w), (10, 12), 1, 0), what="dgrad cs20")


@pytest.mark.parametrize("N,H,W", [(2, 32, 48), (1, 37, 29), (1, 8, 6), (2, 24, 64), (1, 10, 128), (3, 7, 192)])
def test_stem_conv_and_wgrad(ops, N, H, W):
    p = torch.zeros(N, H, W, 4)
    p[..., :3] = rnd(N, H, W, 3, seed=9)
    w = cl(rnd(64, 3, 7, 7, seed=10, scale=0.1))
    wp = ops.pack_stem_weight(cl(w.to(DEV)))
    close(wp, E.pack_stem_weight(w), 0.0, "pack stem")
    y, sums = ops.stem_conv(p.to(DEV), wp, want_stats=True)
    yref = E.stem_conv(p, E.pack_stem_weight(w))
    close(y, yref, what="stem fwd")
    close(sums[0, 0], E.colsum(yref.reshape(-1, 64), moments=True)[0, 0], 1e-5, "stem fused BN statistics: mean")
    close(sums[0, 1], E.colsum(yref.reshape(-1, 64), moments=True)[0, 1], 1e-5, "stem fused BN statistics: variance")
    dy = rnd(*yref.shape, seed=11)
    dwp = torch.empty(64, 7, 8, 4, device=DEV)
    ops.stem_wgrad(p.to(DEV), dy.to(DEV), dwp, False)
    dref = torch.empty(64, 7, 8, 4)
    E.stem_wgrad(p, dy, dref, False)
    got = ops.unpack_stem_weight(dwp, cl(w.to(DEV)))
    close(got, E.unpack_stem_weight(dref, w), 5e-5, "stem wgrad")


@pytest.mark.parametrize("N,H,W,Cc", [(2, 24, 40, 64), (4, 64, 128, 128), (3, 17, 23, 256)])
def test_byte_relu_mask_beside_a_residual_output(ops, monkeypatch, N, H, W, Cc):
    """dcs_bn_act's mask8 (four bits per float4 of a residual block's output) and its two readers -- dcs_bn_bwd_apply and
    the BatchNorm-backward epilogue of a data gradient (relu = 2): every result bitwise the float-mask route's."""
    ops.new_step(True)
    y, r = rnd(N, H, W, Cc, seed=231).to(DEV), rnd(N, H, W, Cc, seed=232).to(DEV)
    gam, bet = (rnd(Cc, seed=233) * 0.1 + 1).to(DEV), (rnd(Cc, seed=234) * 0.1).to(DEV)
    bn = ops.bn_finalize(ops.colsum(y.reshape(-1, Cc), moments=True), gam, bet, torch.zeros(Cc, device=DEV),
                         torch.ones(Cc, device=DEV), N * H * W, True)
    out = ops.bn_act(y, bn, r=r, relu=True)
    m8 = out._mask8
    bits = (out.reshape(-1, 4) > 0).to(torch.uint8)
    assert torch.equal(m8, bits[:, 0] | (bits[:, 1] << 1) | (bits[:, 2] << 2) | (bits[:, 3] << 3))
    g = rnd(N, H, W, Cc, seed=235).to(DEV)
    w = cl(rnd(Cc, Cc, 3, 3, seed=236, scale=0.05).to(DEV))
    dy = rnd(N, H, W, Cc, seed=237).to(DEV) * 1e-2
    # relu | 4: the data gradient's epilogue stores the MASKED gradient and bn_bwd(want_gm=True) hands that tensor back as gm
    wpk0 = ops.pack_dgrad_weight(w)
    gxm, sm = ops.conv_dgrad(dy, wpk0, (H, W), 1, 1, bnb=(y, out, bn, False))
    monkeypatch.setenv("DCS_STORE_MASKED", "0")
    gxr, sr = ops.conv_dgrad(dy, wpk0, (H, W), 1, 1, bnb=(y, out, bn, False))
    monkeypatch.delenv("DCS_STORE_MASKED")
    assert (sm is None and sr is None) or torch.equal(sm, sr)
    if sm is not None:
        assert gxm._dcs_masked is out and not hasattr(gxr, "_dcs_masked")
        assert torch.equal(gxm, gxr * (out > 0))
        d_m, gm_m = ops.bn_bwd(gxm, y, bn, gam, masksrc=out, want_gm=True, sums=sm)
        d_r, gm_r = ops.bn_bwd(gxr, y, bn, gam, masksrc=out, want_gm=True, sums=sr)
        assert gm_m is gxm and gm_r is not gxr and torch.equal(gm_m, gm_r) and torch.equal(d_m, d_r)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("DCS_MASK8", on)
        dyo, _ = ops.bn_bwd(g, y, bn, gam, masksrc=out)
        dg, db = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
        gx, sums = ops.conv_dgrad(dy, ops.pack_dgrad_weight(w), (H, W), 1, 1, bnb=(y, out, bn, False))
        dy2, _ = ops.bn_bwd(gx, y, bn, gam, masksrc=out, dgamma=dg, dbeta=db, sums=sums)
        res[on] = (dyo, gx * (out > 0), sums, dy2, dg, db)      # (with the byte mask the launch stores gx already masked)
    for a, b in zip(res["1"], res["0"]):
        assert (a is None and b is None) or torch.equal(a, b)      # (a split-K launch of a tiny map carries no sums)


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s,dil", [
    (2, 20, 36, 256, 128, 1, 1, 1), (4, 16, 24, 64, 128, 3, 2, 1), (2, 33, 29, 128, 80, 3, 1, 1), (2, 24, 40, 128, 64, 3, 1, 2),
    (8, 64, 128, 64, 64, 1, 1, 1), (1, 8, 8, 512, 512, 3, 1, 1),
])
def test_per_tap_kernel_with_weight_fragments_from_global_is_bitwise_the_lds_staged_one(ops, monkeypatch, N, H, W, Cin, Cout,
                                                                                       k, s, dil):
    """conv_gather_x3_kernel<.., 2, WF = true> (weight fragments straight from global memory, DCS_ACC_WFRAG) against the
    LDS-staged fp16 kernel: 1x1, stride 2 (forward and the parity classes of its data gradient), ragged widths / channel
    counts, dilation, 256-pixel tiles, K splits of a tiny map -- the same products in the same order."""
    ops.new_step(True)
    pad = dil * (k // 2)
    x = rnd(N, H, W, Cin, seed=221).to(DEV)
    w = cl(rnd(Cout, Cin, k, k, seed=222, scale=0.05).to(DEV))
    OH, OW = ops.out_size_d(H, k, s, pad, dil), ops.out_size_d(W, k, s, pad, dil)
    dy = rnd(N, OH, OW, Cout, seed=223).to(DEV) * 1e-3
    ops.tag_max(dy)
    wpk = ops.pack_dgrad_weight(w)
    outs = {}
    for wf in ("1", "0"):
        monkeypatch.setenv("DCS_TAP_WFRAG", wf)
        y, st = ops.conv_fwd(x, w, s, pad, want_stats=True, dil=dil)
        gx = ops.conv_dgrad(dy, wpk, (H, W), s, pad, dil=dil)
        outs[wf] = (y, st, gx)
    for a, b in zip(outs["1"], outs["0"]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(8, 64, 128, 64, 64), (16, 32, 64, 128, 128), (8, 32, 64, 512, 256), (8, 64, 128, 96, 64), (8, 64, 128, 48, 64),
                                            (8, 64, 128, 64, 48)])
def test_double_buffered_halo_kernel_is_bitwise_the_single_buffered_one(ops, libopt, N, H, W, Cin, Cout):
    """conv3x3_x3w_db_kernel (halo of the next 16 channels written into a second LDS buffer behind the MFMAs, one barrier per
    chunk) against conv3x3_x3w_kernel<.., 2>: forward with fused statistics and BatchNorm + ReLU prologue, data gradient with
    the BatchNorm-backward epilogue and a per-tensor scale -- same products in the same order, so every bit agrees."""
    if Cin % 32:                       # an ODD number of 16-channel chunks: the host never sends those to this kernel
        ops.geom_fwd(N, H, W, Cin, Cout, 3, 3, 1, 1)._x3w = True          # (x3w_ok wants K % 32 == 0); the C ABI allows them
        for gd in ops.geoms_dgrad(N, H, W, Cin, Cout, 3, 3, 1, 1, 1):
            gd._x3w = True
    assert ops.x3w_ok(ops.geom_fwd(N, H, W, Cin, Cout, 3, 3, 1, 1))
    ops.new_step(True)
    x = rnd(N, H, W, Cin, seed=201).to(DEV)
    w = cl(rnd(Cout, Cin, 3, 3, seed=202, scale=0.05).to(DEV))
    gi, bi_ = (rnd(Cin, seed=203) * 0.1 + 1).to(DEV), (rnd(Cin, seed=204) * 0.1).to(DEV)
    bn_i = ops.bn_finalize(ops.colsum(x.reshape(-1, Cin), moments=True), gi, bi_, torch.zeros(Cin, device=DEV),
                           torch.ones(Cin, device=DEV), N * H * W, True)
    dy = rnd(N, H, W, Cout, seed=205).to(DEV) * 1e-4
    ops.tag_max(dy)
    wpk = ops.pack_dgrad_weight(w)
    outs = {}
    for db in (1, 0):
        libopt("x3w_db", db)
        y, st = ops.conv_fwd(x, w, 1, 1, want_stats=True, pro=bn_i)
        gx, sums = ops.conv_dgrad(dy, wpk, (H, W), 1, 1, bnb=(x, None, bn_i, True))
        outs[db] = (y, st, gx, sums)
    for a, b in zip(outs[1], outs[0]):
        assert torch.equal(a, b)
    act = E.bn_act(x.cpu(), bn_i.cpu(), relu=True)
    close(outs[1][0], E.conv_fwd(act, w.cpu(), 1, 1), what="double-buffered kernel vs the CPU statement")


@pytest.mark.parametrize("N,H,W", [(8, 128, 256), (6, 160, 320), (16, 95, 191), (10, 112, 256)])
def test_stem_forward_with_the_patch_in_lds(ops, monkeypatch, N, H, W):
    """stem7_h2_kernel (7x7 / stride 2 forward, input patch resident in LDS, fp16 two-piece products) against a float64
    convolution: error <= 1.5x the exact-fp32 MFMA kernel's, fused BatchNorm statistics, image borders (odd sizes: the last
    input row / column is missing); bitwise the per-tap fp16 kernel (DCS_STEM7=0), whose sums it forms in the same order."""
    ops.new_step(True)
    p = torch.zeros(N, H, W, 4)
    p[..., :3] = rnd(N, H, W, 3, seed=191) * 2.5
    w = cl(rnd(64, 3, 7, 7, seed=192, scale=0.1))
    wp = ops.pack_stem_weight(cl(w.to(DEV)))
    pd = p.to(DEV)
    g = ops.geom_stem_fwd(N, H, W)
    if (H + 1) // 2 % 8 or (W + 1) // 2 % 32:
        assert not ops.stem7_ok(g)                       # 48 x 96 outputs: not a multiple of the 8 x 32 tile -> per-tap kernel
        return
    assert ops.stem7_ok(g)
    ref = torch.nn.functional.conv2d(p[..., :3].permute(0, 3, 1, 2).double(), w.double(), None, 2, 3).permute(0, 2, 3, 1)
    y, sums = ops.stem_conv(pd, wp, want_stats=True)
    y2 = ops.stem_conv(pd, wp)
    monkeypatch.setenv("DCS_STEM7", "0")
    ytap = ops.stem_conv(pd, wp)
    monkeypatch.setenv("DCS_CONV_X3", "0")
    y32 = ops.stem_conv(pd, wp)
    monkeypatch.delenv("DCS_CONV_X3"); monkeypatch.delenv("DCS_STEM7")
    assert torch.equal(y, y2)
    scale = float(ref.abs().max())
    e_p, e_t, e_32 = (float((t.cpu().double() - ref).abs().max()) / scale for t in (y, ytap, y32))
    err = y.cpu().double() - ref
    bal = float(err.sum() / err.abs().sum())
    print(f"stem fwd {N}x{H}x{W}: max-rel patch kernel {e_p:.3e} per-tap {e_t:.3e} fp32 {e_32:.3e}; sign balance {bal:+.3f}; "
          f"bitwise the per-tap kernel: {torch.equal(y, ytap)}")
    assert e_p <= 1.5 * e_32 + 1e-7 and abs(bal) < 0.05
    assert torch.equal(y, ytap)                 # same products in the same order as the per-tap fp16 kernel: bitwise
    mom = E.colsum(ref.reshape(-1, 64).float(), moments=True)
    close(sums[0, 0], mom[0, 0], 1e-5, "stem fused BN statistics: mean")
    close(sums[0, 1], mom[0, 1], 1e-5, "stem fused BN statistics: variance")


def test_linear_and_transpose(ops):
    x, w, b = rnd(37, 128, seed=12), rnd(128, 128, seed=13, scale=0.1), rnd(128, seed=14)
    close(ops.linear(x.to(DEV), w.to(DEV), b.to(DEV)), E.linear(x, w, b), what="linear")
    close(ops.transpose(x.to(DEV)), x.t(), 0.0, "transpose")
    dy = rnd(37, 128, seed=15)
    dw = torch.empty(128, 128, device=DEV)
    ops.linear_wgrad(x.to(DEV), dy.to(DEV), dw)
    close(dw, dy.t() @ x, 5e-5, "linear wgrad")


@pytest.mark.parametrize("rows,C,B", [(1000, 64, 1), (77, 128, 1), (4096, 512, 1), (600, 20, 1), (300, 128, 4)])
def test_colsum(ops, rows, C, B):
    x = rnd(B * rows, C, seed=16) + 0.5
    close(ops.colsum(x.to(DEV), B=B, scale=0.5), E.colsum(x, B=B, scale=0.5), 1e-5, "colsum")


def test_batchnorm_forward_backward(ops):
    C, rows = 64, 2 * 9 * 7
    y = rnd(2, 9, 7, C, seed=17) * 2 + 1
    gamma, beta = rnd(C, seed=18) * 0.1 + 1, rnd(C, seed=19) * 0.1
    rm, rv = rnd(C, seed=20) * 0.05, rnd(C, seed=21).abs() + 1
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    rm_c, rv_c = rm.clone(), rv.clone()
    sums = ops.colsum(y.to(DEV).reshape(-1, C))
    bn = ops.bn_finalize(sums, gamma.to(DEV), beta.to(DEV), rm_d, rv_d, rows, True)
    bn_ref = E.bn_finalize(E.colsum(y.reshape(-1, C)), gamma, beta, rm_c, rv_c, rows, True)
    close(bn, bn_ref, 2e-5, "bn params")
    close(rm_d, rm_c, 2e-6, "running mean"); close(rv_d, rv_c, 2e-5, "running var")
    ops.bn_ema_again(bn, rm_d, rv_d, rows)
    E.bn_ema_again(bn_ref, rm_c, rv_c, rows)
    close(rm_d, rm_c, 2e-6, "running mean 2"); close(rv_d, rv_c, 2e-5, "running var 2")
    bn_eval = ops.bn_finalize(None, gamma.to(DEV), beta.to(DEV), rm_d, rv_d, rows, False)
    close(bn_eval, E.bn_finalize(None, gamma, beta, rm_c, rv_c, rows, False), 2e-5, "bn eval")
    r = rnd(2, 9, 7, C, seed=22)
    bn2_ref = E.bn_finalize(E.colsum(r.reshape(-1, C)), gamma, beta, rm_c.clone(), rv_c.clone(), rows, True)
    yd, rd, bnd, bn2d = y.to(DEV), r.to(DEV), bn_ref.to(DEV), bn2_ref.to(DEV)
    for kw_ref, kw in [(dict(), dict()), (dict(r=r), dict(r=rd)), (dict(r=r, bn2=bn2_ref), dict(r=rd, bn2=bn2d)),
                       (dict(relu=False), dict(relu=False))]:
        close(ops.bn_act(yd, bnd, **kw), E.bn_act(y, bn_ref, **kw_ref), 2e-6, f"bn_act {list(kw)}")
    g = rnd(2, 9, 7, C, seed=23)
    out = E.bn_act(y, bn_ref, r=r)
    for name, kw_ref, kw in [("relu", dict(relu=True), dict(relu=True)),
                             ("masksrc", dict(masksrc=out, want_gm=True), dict(masksrc=out.to(DEV), want_gm=True)),
                             ("plain", dict(), dict())]:
        dg_r, db_r = torch.zeros(C), torch.zeros(C)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dy_r, gm_r = E.bn_bwd(g, y, bn_ref, gamma, dgamma=dg_r, dbeta=db_r, **kw_ref)
        dy, gm = ops.bn_bwd(g.to(DEV), yd, bnd, gamma.to(DEV), dgamma=dg, dbeta=db, **kw)
        close(dy, dy_r, 2e-5, f"bn_bwd dy {name}")
        close(dg, dg_r, 2e-5, f"dgamma {name}"); close(db, db_r, 2e-5, f"dbeta {name}")
        if gm_r is not None:
            close(gm, gm_r, 0.0, "gm")
    # accumulate into a slice + accumulate params
    base = rnd(4, 9, 7, C, seed=24)
    acc = base.to(DEV).clone()
    dg = torch.ones(C, device=DEV); db = torch.ones(C, device=DEV)
    ops.bn_bwd(g.to(DEV), yd, bnd, gamma.to(DEV), relu=True, dy_out=acc[:2], acc_dy=True, dgamma=dg, dbeta=db, acc_param=True)
    dg_r, db_r = torch.ones(C), torch.ones(C)
    dy_r, _ = E.bn_bwd(g, y, bn_ref, gamma, relu=True, dgamma=dg_r, dbeta=db_r, acc_param=True)
    close(acc[:2], base[:2] + dy_r, 2e-5, "acc dy"); close(acc[2:], base[2:], 0.0, "untouched")
    close(dg, dg_r, 2e-5, "acc dgamma")


@pytest.mark.parametrize("N,H,W", [(2, 32, 64), (1, 37, 50), (1, 9, 10)])
def test_normalize_pyramid(ops, N, H, W):
    img = rnd(N, 3, H, W, seed=25).abs() * 100
    mean, std = torch.tensor([73.15, 82.90, 72.3]), torch.tensor([47.67, 48.49, 47.73])
    got = ops.normalize_pyramid(img.to(DEV), mean.to(DEV), std.to(DEV))
    ref = E.normalize_pyramid(img, mean, std)
    for a, b in zip(got, ref):
        close(a, b, 2e-6, "pyramid level")


@pytest.mark.parametrize("N,H,W", [(2, 16, 24), (1, 17, 9), (1, 3, 3)])
def test_bn_relu_maxpool_fwd_bwd(ops, monkeypatch, N, H, W):
    C = 64
    y = rnd(N, H, W, C, seed=26)
    bn = torch.stack([rnd(C, seed=27), rnd(C, seed=28) * 0.3, torch.zeros(C), torch.ones(C)])
    out, idx = ops.bn_relu_maxpool(y.to(DEV), bn.to(DEV))
    oref, iref = E.bn_relu_maxpool(y, bn)
    close(out, oref, 1e-6, "pool fwd")
    g = rnd(*oref.shape, seed=29) * (oref > 0)      # positions with a zero maximum are ties; relu kills them anyway
    close(ops.maxpool_bwd(g.to(DEV), idx, H, W), E.maxpool_bwd(g, iref, H, W), 1e-6, "pool bwd")
    # fused stem backward (pool adjoint gathered inside both BatchNorm passes) vs the two-step statement
    bn[2], bn[3] = rnd(C, seed=30) * 0.1, 1.0 + 0.2 * rnd(C, seed=31).abs()
    gamma = 1.0 + 0.1 * rnd(C, seed=32)
    g2 = rnd(*oref.shape, seed=33)
    # sums: from the pooled tensors alone (dcs_bn_pool_bwd_partial_pooled; the random `scale` row has channels on either
    # side of its |gamma| threshold, so both the z route and the argmax fetch run) and from the un-pooled map
    assert idx._pooled is out
    for pooled in ("1", "0"):
        monkeypatch.setenv("DCS_POOL_SUMS", pooled)
        for training, acc in ((True, False), (False, True)):
            dg0, db0 = rnd(C, seed=34), rnd(C, seed=35)
            dgd, dbd = dg0.to(DEV).clone(), db0.to(DEV).clone()
            dy = ops.bn_pool_bwd(g2.to(DEV), idx, y.to(DEV), bn.to(DEV), gamma.to(DEV), dgamma=dgd, dbeta=dbd,
                                 acc_param=acc, training=training)
            dgr, dbr = dg0.clone(), db0.clone()
            ref = E.bn_pool_bwd(g2, iref, y, bn, gamma, dgamma=dgr, dbeta=dbr, acc_param=acc, training=training)
            close(dy, ref, 2e-5, f"fused pool+bn bwd training={training} pooled sums={pooled}")
            close(dgd, dgr, 2e-5, "fused pool+bn dgamma"); close(dbd, dbr, 2e-5, "fused pool+bn dbeta")


@pytest.mark.parametrize("IH,IW,OH,OW", [(4, 8, 8, 16), (3, 5, 6, 10), (2, 3, 3, 5), (1, 2, 2, 3), (5, 7, 9, 12), (6, 9, 24, 36),
                                         (5, 4, 10, 16)])
def test_upsample_add_and_adjoint(ops, IH, IW, OH, OW):
    N, C = 2, 128
    x = rnd(N, IH, IW, C, seed=30)
    sk = [rnd(N, OH, OW, C, seed=31 + i) for i in range(3)]
    for n in (1, 2, 3):
        close(ops.upsample_add(x.to(DEV), [s.to(DEV) for s in sk[:n]], OH, OW), E.upsample_add(x, sk[:n], OH, OW), 2e-6,
              f"upsample_add {n}")
    g = rnd(N, OH, OW, C, seed=35)
    gx = ops.upsample_bwd(g.to(DEV), IH, IW)
    close(gx, E.upsample_bwd(g, IH, IW), 1e-5, "upsample bwd")
    if ops.x2h_on():                   # the maximum word the fp16 two-piece consumers scale by
        assert int(gx._dcs_max.item()) == int(gx.abs().max().view(torch.int32).item())
    base = rnd(N, IH, IW, C, seed=36)
    acc = base.to(DEV).clone()
    ops.upsample_bwd(g.to(DEV), IH, IW, out=acc, accumulate=True)
    close(acc, base + E.upsample_bwd(g, IH, IW), 1e-5, "upsample bwd acc")


@pytest.mark.parametrize("IH,IW,OH,OW", [(8, 16, 32, 64), (30, 50, 120, 200), (7, 5, 27, 21)])
def test_logits_upsample_and_adjoint(ops, IH, IW, OH, OW):
    x = torch.zeros(2, IH, IW, 20)
    x[..., :19] = rnd(2, IH, IW, 19, seed=37)
    close(ops.upsample_to_nchw(x.to(DEV), 19, OH, OW), E.upsample_to_nchw(x, 19, OH, OW), 2e-6, "logits up")
    g = rnd(2, 19, OH, OW, seed=38)
    close(ops.upsample_to_nchw_bwd(g.to(DEV), IH, IW, 20), E.upsample_to_nchw_bwd(g, IH, IW, 20), 1e-5, "logits up bwd")
    gs = torch.tensor([0.37])
    close(ops.upsample_to_nchw_bwd(g.to(DEV), IH, IW, 20, gscale=gs.to(DEV)), 0.37 * E.upsample_to_nchw_bwd(g, IH, IW, 20),
          1e-5, "logits up bwd scaled")
    g40 = rnd(1, 35, OH, OW, seed=39)          # > 32 classes: single-pass gather kernel
    close(ops.upsample_to_nchw_bwd(g40.to(DEV), IH, IW, 36), E.upsample_to_nchw_bwd(g40, IH, IW, 36), 1e-5, "logits up bwd 35")


@pytest.mark.parametrize("mode", ["full", "plain_focal", "no_class_weights", "no_EDT", "ce"])
def test_seg_loss(ops, mode):
    g = np.random.default_rng(39)
    lg = rnd(2, 19, 24, 40, seed=40) * 2
    tgt = torch.from_numpy(g.integers(0, 19, size=(2, 24, 40)).astype(np.int64))
    tgt[:, :2, :] = 255
    ldw = torch.from_numpy(g.random((2, 24, 40)).astype(np.float32))
    ldw[tgt == 255] = 0
    cw = torch.from_numpy((1.0 / np.log(1.1 + g.random(19) * 0.2)).astype(np.float32))
    t_ref, t_dev = tgt.clone(), tgt.clone().to(DEV)
    args = (None, None) if mode == "ce" else (ldw, cw)
    out_r, grad_r = E.seg_loss(lg, t_ref, *args, mode)
    out, grad = ops.seg_loss(lg.to(DEV), t_dev, *(a.to(DEV) if a is not None else None for a in args), mode)
    close(out, out_r, 1e-5, "loss/count")
    close(grad, grad_r, 1e-5, "grad")
    assert torch.equal(t_dev.cpu(), t_ref)                      # in-place 255 -> 0 for focal modes only
    ops.scale_inplace(grad, torch.tensor([1.2], device=DEV), out[2:3])
    close(grad, grad_r * 1.2 * out_r[2], 1e-5, "scaled grad")


@pytest.mark.parametrize("mode", ["full", "plain_focal", "no_class_weights", "no_EDT", "ce"])
@pytest.mark.parametrize("N,ih,iw,F_", [(2, 13, 37, 4), (1, 8, 32, 4), (3, 50, 82, 4), (2, 9, 33, 2)])
def test_seg_loss_fused(ops, mode, N, ih, iw, F_):
    """dcs_seg_loss_fused (upsampling + log-softmax + focal/CE + adjoint of the upsampling in one kernel, nothing at full
    resolution in memory) against upsample_to_nchw -> seg_loss -> upsample_to_nchw_bwd: partial tiles, image borders,
    ignore pixels, the in-place 255 -> 0 rewrite, x2 and x4."""
    if mode != "full" and (N, ih) != (2, 13):
        pytest.skip("one geometry per non-default mode")
    g = np.random.default_rng(61 + ih)
    H, W = F_ * ih, F_ * iw
    lr = torch.zeros(N, ih, iw, 20)
    lr[..., :19] = rnd(N, ih, iw, 19, seed=62 + ih) * 2
    lr[..., 19] = 7.0                                            # the pad channel must be ignored
    tgt = torch.from_numpy(g.integers(0, 19, size=(N, H, W)).astype(np.int64))
    tgt[:, :3, :] = 255
    tgt[:, :, -2:] = 255
    ldw = torch.from_numpy(g.random((N, H, W)).astype(np.float32))
    ldw[tgt == 255] = 0
    cw = torch.from_numpy((1.0 / np.log(1.1 + g.random(19) * 0.2)).astype(np.float32))
    t_ref, t_dev = tgt.clone(), tgt.clone().to(DEV)
    args = (None, None) if mode == "ce" else (ldw, cw)
    out_r, grad_r = E.seg_loss_fused(lr.double(), 19, t_ref, *(a.double() if a is not None else None for a in args), mode)
    dargs = tuple(a.to(DEV) if a is not None else None for a in args)
    out, grad = ops.seg_loss_fused(lr.to(DEV), 19, t_dev, *dargs, mode)
    close(out, out_r, 2e-6, "loss / count / 1/count")
    close(grad[..., :19], grad_r[..., :19], 2e-5, "gradient of the low-resolution logits")
    assert float(grad[..., 19].abs().max()) == 0.0
    assert torch.equal(t_dev.cpu(), t_ref)                      # in-place 255 -> 0 for focal modes only
    # the unfused kernels evaluate the same interpolation weights: same loss to rounding, and bitwise reproducible
    t2 = tgt.clone().to(DEV)
    up = ops.upsample_to_nchw(lr.to(DEV), 19, H, W)
    out_u, grad_u = ops.seg_loss(up, t2, *dargs, mode)
    close(out, out_u.cpu(), 1e-6, "fused vs unfused loss")
    close(grad[..., :19], ops.upsample_to_nchw_bwd(grad_u, ih, iw, 20)[..., :19].cpu(), 2e-5, "fused vs unfused gradient")
    out2, grad2 = ops.seg_loss_fused(lr.to(DEV), 19, tgt.clone().to(DEV), *dargs, mode)
    assert torch.equal(out, out2) and torch.equal(grad, grad2)


def test_seg_loss_no_valid_pixels(ops):
    lg = rnd(1, 19, 4, 4, seed=41)
    t = torch.full((1, 4, 4), 255, dtype=torch.int64, device=DEV)
    out, grad = ops.seg_loss(lg.to(DEV), t, torch.zeros(1, 4, 4, device=DEV), torch.ones(19, device=DEV), "full")
    assert out.cpu().tolist() == [0.0, 0.0, 0.0] and int(t.max()) == 0


@pytest.mark.parametrize("h,w,scale", [(12, 20, 4), (33, 47, 4), (64, 64, 2)])
def test_anchor_keys_and_select(ops, h, w, scale):
    g = np.random.default_rng(42)
    N, C = 3, 19
    lg = torch.zeros(N, h, w, 20)
    lg[..., :19] = rnd(N, h, w, 19, seed=43)
    labels = torch.from_numpy(g.integers(0, 7, size=(N, h * scale, w * scale)).astype(np.int64))
    labels[:, :5, :] = 255
    key_r, hist_r = E.anchor_keys_raw(lg, N, h, w, 20, C, labels)
    key, hist = ops.anchor_keys_raw(lg.to(DEV), N, h, w, 20, C, labels.to(DEV))
    assert torch.equal(key.cpu(), key_r) and torch.equal(hist.cpu(), hist_r)
    counts = hist_r.sum(1)
    req = []
    for n in range(N):
        for k in range(2 * C):
            c = int(counts[n, k])
            for r in {0, c // 2, c - 1, c}:
                if r >= 0:
                    req.append([n, k, r])
    req = torch.tensor(req, dtype=torch.int32)
    sel = ops.anchor_select(key, hist, req.to(DEV), C)
    assert torch.equal(sel.cpu(), E.anchor_select(key_r, hist_r, req, C))


def test_gather_scatter_rows(ops):
    feat = rnd(500, 128, seed=44)
    idx = torch.from_numpy(np.random.default_rng(45).permutation(500)[:37].astype(np.int32))
    X = ops.gather_rows(feat.to(DEV), idx.to(DEV))
    close(X, feat[idx.long()], 0.0, "gather")
    gf = rnd(500, 128, seed=46)
    gfd = gf.to(DEV).clone()
    ops.scatter_add_rows(X, idx.to(DEV), gfd)
    ref = gf.clone()
    ref[idx.long()] += feat[idx.long()]
    close(gfd, ref, 1e-7, "scatter")


def _contrast_case(A, mode, C=128, seed=0):
    g = np.random.default_rng(47 + A + seed)
    X = rnd(A, C, seed=48 + A + seed) * (0.7 if C == 128 else 0.2)
    if mode == 0:
        T = A // 2
        y = torch.from_numpy(np.tile(g.integers(0, 6, size=T), 2).astype(np.float32)) if A % 2 == 0 else \
            torch.from_numpy(g.integers(0, 3, size=A).astype(np.float32))
    else:
        y = torch.from_numpy(np.tile(g.integers(0, 3, size=A // 2), 2).astype(np.float32))
    return X, y


@pytest.mark.parametrize("A,mode", [(8, 1), (64, 1), (76, 0), (304, 0), (608, 0), (33, 0), (1000, 0), (256, 1),
                                    (1216, 0), (2437, 0), (4864, 0), (1280, 1)])   # > 1024: the symmetric strip kernels
def test_contrast_fwd_bwd(ops, A, mode):
    """Fused similarity / InfoNCE kernels (dcs_contrast_fused) against the float64 autograd statement of
    utils/loss.py:339-389 / :175-204.  A <= 1024: strip-resident family; larger: symmetric tile sweeps."""
    X, y = _contrast_case(A, mode)
    loss_r, dX_r = E.contrast_fwd_bwd(X.double(), y.double(), mode)
    loss, dX = ops.contrast_fwd_bwd(X.to(DEV), y.to(DEV), mode)
    close(loss, loss_r, 2e-5, "loss")
    close(dX, dX_r, 2e-4, "dX")
    loss2, dX2 = ops.contrast_fwd_bwd(X.to(DEV), y.to(DEV), mode)          # deterministic (no float atomics)
    assert torch.equal(loss, loss2) and torch.equal(dX, dX2)


@pytest.mark.parametrize("A,mode", [(76, 0), (608, 0), (1000, 0), (64, 1), (256, 1)])
def test_contrast_small_family_one_launch_equals_two(ops, libopt, A, mode):
    """A <= 1024: the cooperative single launch (statistics, grid barrier, gradient sweep reading the S strip back from
    LDS) performs the arithmetic of the two-launch form (which recomputes the same S tiles): bitwise the same loss and dX."""
    X, y = _contrast_case(A, mode)
    libopt("contrast_fused", 1)
    l1, d1 = ops.contrast_fwd_bwd(X.to(DEV), y.to(DEV), mode)
    libopt("contrast_fused", 0)
    l0, d0 = ops.contrast_fwd_bwd(X.to(DEV), y.to(DEV), mode)
    assert torch.equal(l0, l1) and torch.equal(d0, d1)


@pytest.mark.parametrize("A,cap,world,mode", [(37, 64, 2, 0), (300, 608, 2, 0), (10, 16, 4, 1), (500, 608, 4, 0)])
def test_contrast_padded_strided_gather_buffer(ops, A, cap, world, mode):
    """The data-parallel layout: [world * cap] rows of 132 floats (128 channels, label, 3 pad), label -1 on the rows a
    rank did not fill.  Padding rows must take part in nothing: loss and gradient equal the compact problem's, and the
    padding rows' gradient is exactly zero."""
    g = np.random.default_rng(5 + A)
    counts = [A] + [int(v) for v in g.integers(0 if mode == 0 else 2, cap + 1, size=world - 1)]
    if mode == 1:
        counts = [c - c % 2 for c in counts]
    buf = torch.zeros(world * cap, 132)
    buf[:, 128] = -1.0
    Xs, ys = [], []
    for r, c in enumerate(counts):
        Xr, yr = _contrast_case(max(c, 2), mode, seed=r)
        Xr, yr = Xr[:c], yr[:c]
        buf[r * cap:r * cap + c, :128] = Xr
        buf[r * cap:r * cap + c, 128] = yr
        Xs.append(Xr); ys.append(yr)
    Xc, yc = torch.cat(Xs), torch.cat(ys)
    loss_r, dX_r = E.contrast_fwd_bwd(Xc.double(), yc.double(), mode)
    bd = buf.to(DEV)
    loss, dX = ops.contrast_fwd_bwd(bd[:, :128], bd[:, 128], mode)
    close(loss, loss_r, 2e-5, "loss")
    dX = dX.cpu()
    valid = buf[:, 128] >= 0
    close(dX[valid], dX_r, 2e-4, "dX of the valid rows")
    assert float(dX[~valid].abs().max()) == 0.0 if bool((~valid).any()) else True
    # and the dense emulation of the padded problem agrees with itself
    loss_p, dX_p = E.contrast_fwd_bwd(buf[:, :128].double(), buf[:, 128].double(), mode)
    close(loss_p, loss_r, 1e-12, "emulation, padded vs compact")


def test_contrast_explicit_mask_and_wide_features(ops):
    """SupConLoss(mask=...) (utils/loss.py:148-159: an explicit, possibly asymmetric [bsz,bsz] mask) and the 2048-channel
    rows of DeepLab's pixel contrast (the kernel returns G + G^T, one GEMM finishes dX)."""
    g = np.random.default_rng(3)
    b = 12
    X = rnd(2 * b, 128, seed=91) * 0.7
    mask = torch.from_numpy((g.random((b, b)) < 0.4).astype(np.float32))
    mask[torch.arange(b), (torch.arange(b) + 1) % b] = 1.0            # every row keeps a positive
    y = torch.arange(b, dtype=torch.float32).repeat(2)
    loss_r, dX_r = E.contrast_fwd_bwd(X.double(), y.double(), 1, mask=mask.double())
    loss, dX = ops.contrast_fwd_bwd(X.to(DEV), y.to(DEV), 1, mask=mask.to(DEV))
    close(loss, loss_r, 2e-5, "masked supcon loss")
    close(dX, dX_r, 2e-4, "masked supcon dX")
    Xw, yw = _contrast_case(152, 0, C=2048)
    loss_r, dX_r = E.contrast_fwd_bwd(Xw.double(), yw.double(), 0)
    loss, dX = ops.contrast_fwd_bwd(Xw.to(DEV), yw.to(DEV), 0)
    close(loss, loss_r, 2e-5, "wide loss")
    close(dX, dX_r, 2e-4, "wide dX")


def test_adam_and_small_helpers(ops):
    p, gr = cl(rnd(64, 64, 3, 3, seed=50)), cl(rnd(64, 64, 3, 3, seed=51))
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pd, gd, md, vd = (cl(t.to(DEV)) for t in (p, gr, m, v))
    for step in (1, 2, 3):
        E.adam_step(p, gr, m, v, 1e-4, 0.9, 0.99, 1e-8, 2.5e-5, step)
        ops.adam_step(pd, gd, md, vd, 1e-4, 0.9, 0.99, 1e-8, 2.5e-5, step)
    close(pd, p, 1e-6, "adam p"); close(vd, v, 1e-5, "adam v")
    a, b = rnd(1000, seed=52), rnd(1000, seed=53)
    ad = a.to(DEV).clone()
    ops.axpy(ad, b.to(DEV), 0.25)
    close(ad, a + 0.25 * b, 1e-6, "axpy")
    close(ops.sum_scalar(a.to(DEV), 0.5), a.double().sum().float().reshape(1) * 0.5, 1e-5, "sum")
    close(ops.relu_bwd(a.to(DEV), b.to(DEV)), a * (b > 0), 0.0, "relu bwd")
    gt, vv = rnd(2, 3, 5, 128, seed=54), rnd(2, 128, seed=55)
    gd = gt.to(DEV).clone()
    ops.add_rowvec_bcast(gd, vv.to(DEV), 0.1)
    close(gd, gt + 0.1 * vv.view(2, 1, 1, 128), 1e-6, "bcast")
    gw = torch.full((2, 3, 5, 128), float("nan"), device=DEV)
    ops.add_rowvec_bcast(gw, vv.to(DEV), 0.1, accumulate=False)
    close(gw, (0.1 * vv.view(2, 1, 1, 128)).expand(2, 3, 5, 128), 1e-6, "bcast write")


def test_ops_refuse_cpu_tensors(ops):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv_fwd(torch.zeros(1, 4, 4, 64), cl(torch.zeros(64, 64, 3, 3)), 1, 1)


@pytest.mark.parametrize("N,H,W,Cin,Cout,s,dil", [(1, 12, 14, 64, 64, 1, 2), (2, 9, 11, 128, 256, 1, 6), (1, 20, 24, 64, 128, 2, 1),
                                                  (1, 8, 16, 256, 64, 1, 12)])
def test_dilated_conv(ops, N, H, W, Cin, Cout, s, dil):
    x = rnd(N, H, W, Cin, seed=60)
    w = cl(rnd(Cout, Cin, 3, 3, seed=61, scale=0.05))
    y_ref = E.conv_fwd(x, w, s, dil, dil=dil)
    close(ops.conv_fwd(x.to(DEV), cl(w.to(DEV)), s, dil, dil=dil), y_ref, what="dilated fwd")
    dy = rnd(*y_ref.shape, seed=62)
    gx = ops.conv_dgrad(dy.to(DEV), ops.pack_dgrad_weight(cl(w.to(DEV))), (H, W), s, dil, dil=dil)
    close(gx, E.conv_dgrad(dy, E.pack_dgrad_weight(w), (H, W), s, dil, dil=dil), what="dilated dgrad")
    dw = torch.empty_like(cl(w.to(DEV)))
    ops.conv_wgrad(x.to(DEV), dy.to(DEV), dw, s, dil, False, dil=dil)
    dref = torch.empty_like(w)
    E.conv_wgrad(x, dy, dref, s, dil, False, dil=dil)
    close(dw, dref, 5e-5, "dilated wgrad")


@pytest.mark.parametrize("k,W", [(1, 10), (3, 32), (3, 13)])
def test_virtual_concat_convolution(ops, k, W):
    """conv(cat([a, b], C)) == conv(a, w[:, :48]) + conv(b, w[:, 48:]) with slices of ONE weight tensor."""
    N, H, Ca, Cb, Cout = 2, 6, 48, 256, 64
    a, b = rnd(N, H, W, Ca, seed=63), rnd(N, H, W, Cb, seed=64)
    w = cl(rnd(Cout, Ca + Cb, k, k, seed=65, scale=0.05))
    pad = k // 2
    ref = E.conv_fwd(torch.cat([a, b], -1), w, 1, pad)
    wd = cl(w.to(DEV))
    y = ops.conv_fwd(a.to(DEV), wd, 1, pad, koff=0)
    ops.conv_fwd(b.to(DEV), wd, 1, pad, koff=Ca, out=y)
    close(y, ref, what="sliced fwd")
    dy = rnd(*ref.shape, seed=66)
    gref = E.conv_dgrad(dy, E.pack_dgrad_weight(w), (H, W), 1, pad)
    close(ops.conv_dgrad(dy.to(DEV), ops.pack_dgrad_weight(wd, 0, Ca), (H, W), 1, pad), gref[..., :Ca], what="sliced dgrad a")
    close(ops.conv_dgrad(dy.to(DEV), ops.pack_dgrad_weight(wd, Ca, Cb), (H, W), 1, pad), gref[..., Ca:], what="sliced dgrad b")
    dw = torch.full_like(wd, 7.0)
    ops.conv_wgrad(a.to(DEV), dy.to(DEV), dw, 1, pad, False, koff=0)
    ops.conv_wgrad(b.to(DEV), dy.to(DEV), dw, 1, pad, False, koff=Ca)
    dref = torch.empty_like(w)
    E.conv_wgrad(torch.cat([a, b], -1), dy, dref, 1, pad, False)
    close(dw, dref, 5e-5, "sliced wgrad")


def test_dropout(ops):
    x = rnd(4, 8, 16, 256, seed=67)
    noise = (torch.from_numpy(np.random.default_rng(68).random(x.shape).astype(np.float32)) < 0.9).float()
    out, mask = ops.dropout(x.to(DEV), 0.1, noise.to(DEV))
    close(out, x * noise / 0.9, 1e-6, "dropout with host noise")
    assert torch.equal(mask.cpu(), noise.to(torch.uint8))
    g = rnd(*x.shape, seed=69)
    close(ops.dropout_bwd(g.to(DEV), mask, 0.1), g * noise / 0.9, 1e-6, "dropout bwd")
    out2, mask2 = ops.dropout(x.to(DEV), 0.1, None, seed=5)
    keep = float(mask2.float().mean())
    assert 0.88 < keep < 0.92, keep
    close(out2, x * mask2.cpu().float() / 0.9, 1e-6, "dropout with device mask")
    _, mask3 = ops.dropout(x.to(DEV), 0.1, None, seed=6)
    assert not torch.equal(mask2, mask3)


def test_wide_channel_batchnorm_reductions(ops):
    C, rows = 2048, 300
    y = rnd(rows, C, seed=70) + 0.3
    close(ops.colsum(y.to(DEV)), E.colsum(y), 1e-5, "colsum C=2048")
    gamma, beta = rnd(C, seed=71) * 0.1 + 1, rnd(C, seed=72) * 0.1
    bn_ref = E.bn_finalize(E.colsum(y), gamma, beta, torch.zeros(C), torch.ones(C), rows, True)
    g = rnd(rows, C, seed=73)
    dg_r, db_r, dg, db = torch.zeros(C), torch.zeros(C), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dy_r, _ = E.bn_bwd(g, y, bn_ref, gamma, relu=True, dgamma=dg_r, dbeta=db_r)
    dy, _ = ops.bn_bwd(g.to(DEV), y.to(DEV), bn_ref.to(DEV), gamma.to(DEV), relu=True, dgamma=dg, dbeta=db)
    close(dy, dy_r, 2e-5, "bn_bwd C=2048"); close(dg, dg_r, 2e-5, "dgamma C=2048")


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s,masked,acc", [
    (2, 24, 40, 64, 64, 3, 1, False, False),      # conv2 data gradient -> bn1 (ReLU mask recomputed from y)
    (4, 73, 71, 128, 128, 3, 1, True, True),      # conv1 data gradient accumulating into the residual gradient -> previous bn2
    (2, 16, 24, 64, 128, 3, 2, True, True),       # stride 2: four parity-class launches, one set of sums
    (4, 33, 47, 128, 128, 1, 1, True, True),      # skip projection (1x1) as the last writer
    (64, 96, 96, 64, 64, 3, 1, False, False),     # > 2048 row tiles: two-level final reduce
    (4, 128, 256, 128, 128, 3, 1, False, False),  # decoder blend: 128-wide tiles, mask recomputed from y (the shape on which
                                                  # an SLP-vectorised epilogue returned wrong odd-channel sums: csrc/Makefile)
])
def test_conv_dgrad_with_batchnorm_backward_sums(ops, N, H, W, Cin, Cout, k, s, masked, acc):
    """dcs_conv_gather_bnbwd: the data gradient's epilogue also reduces sum(gm) and sum(gm * xhat) of the BatchNorm
    backward that consumes it (gm = final value x ReLU mask).  The gradient itself must be unchanged, the sums must equal
    what the stand-alone reduction pass (dcs_colsum_partial mode 1) computes from the written tensor."""
    pad = k // 2
    OH, OW = E.conv_fwd(torch.zeros(1, H, W, Cin), cl(torch.zeros(Cout, Cin, k, k)), s, pad).shape[1:3]
    dy = rnd(N, OH, OW, Cout, seed=71)
    w = cl(rnd(Cout, Cin, k, k, seed=72, scale=0.05))
    y = rnd(N, H, W, Cin, seed=73) * 1.5 + 0.3
    mask = torch.relu(rnd(N, H, W, Cin, seed=74)) if masked else None
    base = rnd(N, H, W, Cin, seed=75)
    gamma, beta = rnd(Cin, seed=76) * 0.1 + 1, rnd(Cin, seed=77) * 0.1
    yd, dyd, wd = y.to(DEV), dy.to(DEV), cl(w.to(DEV))
    bn = ops.bn_finalize(ops.colsum(yd.reshape(-1, Cin), moments=True), gamma.to(DEV), beta.to(DEV), torch.zeros(Cin, device=DEV),
                         torch.ones(Cin, device=DEV), N * H * W, True)
    wp = ops.pack_dgrad_weight(wd)
    md = mask.to(DEV) if masked else None
    ref_out = base.to(DEV).clone() if acc else None
    ref = ops.conv_dgrad(dyd, wp, (H, W), s, pad, out=ref_out, accumulate=acc)
    out = base.to(DEV).clone() if acc else None
    got, sums = ops.conv_dgrad(dyd, wp, (H, W), s, pad, out=out, accumulate=acc, bnb=(yd, md, bn, not masked))
    assert sums is not None and torch.equal(got, ref)             # (tiny maps run split-K and return sums = None: not these)
    dg, db = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)
    dy_a, _ = ops.bn_bwd(ref, yd, bn, gamma.to(DEV), masksrc=md, relu=not masked, dgamma=dg, dbeta=db)
    dg2, db2 = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)
    dy_b, _ = ops.bn_bwd(got, yd, bn, gamma.to(DEV), masksrc=md, relu=not masked, dgamma=dg2, dbeta=db2, sums=sums)
    close(db2, db.cpu(), 2e-5, "sum gm (dbeta)")
    close(dg2, dg.cpu(), 2e-5, "sum gm * xhat (dgamma)")
    close(dy_b, dy_a.cpu(), 2e-5, "BatchNorm backward with the fused sums")
    got2, sums2 = ops.conv_dgrad(dyd, wp, (H, W), s, pad, out=(base.to(DEV).clone() if acc else None), accumulate=acc,
                                 bnb=(yd, md, bn, not masked))
    assert torch.equal(sums, sums2)                                   # fixed-order reduction: bitwise reproducible


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,bias,cs", [
    (2, 24, 64, 64, 64, 3, False, None),        # conv2 of a BasicBlock; nine-tap weight gradient (W % 32 == 0)
    (2, 37, 29, 128, 128, 3, False, None),      # ragged map: generic weight-gradient kernel, partial row tiles
    (8, 4, 8, 512, 512, 3, False, None),        # deep layer of a small input: split-K forward, K = DCS_PRO_MAXK
    (2, 20, 36, 128, 19, 1, True, 20),          # segmentation head: 1x1, bias, padded logits stride
    (1, 16, 32, 256, 128, 3, False, None),      # wide source, 128-wide tiles
    (2, 9, 11, 48, 64, 3, False, None),         # source width not a multiple of the 32-channel chunk
])
def test_conv_with_batchnorm_relu_prologue(ops, N, H, W, Cin, Cout, k, bias, cs):
    """dcs_conv_gather_pro / dcs_conv_wgrad_pro: the convolution (and its weight gradient) of relu(y * scale + shift)
    taken on the fly must be BITWISE the convolution of the materialised activation (same kernel, same operand values,
    zero in the padding), and equal to the CPU emulation within the fp32 convolution tolerance."""
    pad = k // 2
    y = rnd(N, H, W, Cin, seed=91) * 1.3 + 0.2
    w = cl(rnd(Cout, Cin, k, k, seed=92, scale=0.05))
    b = rnd(Cout, seed=93) if bias else None
    gamma, beta = rnd(Cin, seed=94) * 0.1 + 1, rnd(Cin, seed=95) * 0.1 + 0.05
    yd, wd = y.to(DEV), cl(w.to(DEV))
    bd = b.to(DEV) if bias else None
    bn = ops.bn_finalize(ops.colsum(yd.reshape(-1, Cin), moments=True), gamma.to(DEV), beta.to(DEV), torch.zeros(Cin, device=DEV),
                         torch.ones(Cin, device=DEV), N * H * W, True)
    assert ops.pro_ok(Cin)
    z = ops.bn_act(yd, bn, relu=True)
    stats = not bias
    ref = ops.conv_fwd(z, wd, 1, pad, bias=bd, dst_cs=cs, want_stats=stats)
    got = ops.conv_fwd(yd, wd, 1, pad, bias=bd, dst_cs=cs, want_stats=stats, pro=bn)
    if stats:
        assert torch.equal(got[1], ref[1])
        got, ref = got[0], ref[0]
    assert torch.equal(got, ref)
    emu = E.conv_fwd(y, w, 1, pad, bias=b, dst_cs=cs, pro=bn.cpu())
    close(got, emu, 2e-4, "prologue convolution vs emulation")
    # accumulate form (virtual-concatenation slices)
    acc0 = rnd(*ref.shape, seed=96).to(DEV)
    if not bias:
        a_ref = ops.conv_fwd(z, wd, 1, pad, out=acc0.clone())
        a_got = ops.conv_fwd(yd, wd, 1, pad, out=acc0.clone(), pro=bn)
        assert torch.equal(a_got, a_ref)
    # weight gradient
    dy = rnd(*ref.shape, seed=97).to(DEV)
    if cs:
        dy[..., Cout:] = 0
    dw_ref, dw_got = cl(torch.empty(Cout, Cin, k, k, device=DEV)), cl(torch.empty(Cout, Cin, k, k, device=DEV))
    ops.conv_wgrad(z, dy, dw_ref, 1, pad, False)
    ops.conv_wgrad(yd, dy, dw_got, 1, pad, False, pro=bn)
    assert torch.equal(dw_got, dw_ref)
    dw_emu = cl(torch.empty(Cout, Cin, k, k))
    E.conv_wgrad(y, dy.cpu(), dw_emu, 1, pad, False, pro=bn.cpu())
    close(dw_got, dw_emu, 2e-4, "prologue weight gradient vs emulation")


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s", [
    (2, 24, 40, 64, 64, 3, 1), (1, 33, 29, 128, 128, 3, 1), (2, 16, 24, 64, 128, 3, 2), (1, 20, 36, 256, 128, 1, 1),
    (1, 12, 16, 512, 512, 3, 1), (2, 9, 11, 128, 80, 3, 1),
])
def test_split_bf16_convolution_has_fp32_class_error(ops, monkeypatch, N, H, W, Cin, Cout, k, s):
    """csrc/conv_split.hip (fp32 operands as three bf16 pieces, six bf16 MFMAs per product) against the exact-fp32 MFMA
    kernel, both measured against a float64 convolution of the same fp32 inputs: the split kernel's error may not exceed
    1.5x the fp32 kernel's (+ a 1e-7 floor), for the forward and the data gradient; and it must really be the one that ran
    (the two results differ in the last bits)."""
    pad = k // 2
    x = rnd(N, H, W, Cin, seed=101)
    w = cl(rnd(Cout, Cin, k, k, seed=102, scale=0.05))
    ref = E.conv_fwd(x.double(), w.double(), s, pad)
    xd, wd = x.to(DEV), cl(w.to(DEV))
    g = ops.geom_fwd(N, H, W, Cin, Cout, k, k, s, pad)
    assert ops.x3_ok(g)
    got = ops.conv_fwd(xd, wd, s, pad).cpu()
    monkeypatch.setenv("DCS_CONV_X3", "0")
    f32 = ops.conv_fwd(xd, wd, s, pad).cpu()
    monkeypatch.delenv("DCS_CONV_X3")
    scale = float(ref.abs().max())
    e_x3, e_32 = float((got.double() - ref).abs().max()) / scale, float((f32.double() - ref).abs().max()) / scale
    l_x3 = float((got.double() - ref).norm() / ref.norm()); l_32 = float((f32.double() - ref).norm() / ref.norm())
    print(f"fwd  max-rel x3 {e_x3:.3e} fp32 {e_32:.3e} | l2-rel x3 {l_x3:.3e} fp32 {l_32:.3e}")
    assert e_x3 <= 1.5 * e_32 + 1e-7 and l_x3 <= 1.5 * l_32 + 1e-8
    assert not torch.equal(got, f32)
    # no rounding BIAS: the bf16 MFMA rounds its sums downwards; the kernel alternates the sign of chunk pairs so that
    # this cancels (conv_split.hip).  Signed error sum / absolute error sum: ~N^-1/2 for zero-mean errors, -0.1..-0.3
    # without the alternation.
    err = got.double() - ref
    assert abs(float(err.sum() / err.abs().sum())) < 0.03, float(err.sum() / err.abs().sum())
    # data gradient
    OH, OW = ref.shape[1:3]
    dy = rnd(N, OH, OW, Cout, seed=103)
    refd = E.conv_dgrad(dy.double(), E.pack_dgrad_weight(w.double()), (H, W), s, pad)
    wp = ops.pack_dgrad_weight(wd)
    gotd = ops.conv_dgrad(dy.to(DEV), wp, (H, W), s, pad).cpu()
    monkeypatch.setenv("DCS_CONV_X3", "0")
    f32d = ops.conv_dgrad(dy.to(DEV), wp, (H, W), s, pad).cpu()
    monkeypatch.delenv("DCS_CONV_X3")
    scale = float(refd.abs().max())
    e_x3, e_32 = float((gotd.double() - refd).abs().max()) / scale, float((f32d.double() - refd).abs().max()) / scale
    print(f"dgrad max-rel x3 {e_x3:.3e} fp32 {e_32:.3e}")
    assert e_x3 <= 1.5 * e_32 + 1e-7


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 64, 64, 64), (4, 16, 64, 128, 128), (2, 8, 96, 256, 128), (2, 8, 64, 128, 256),
                                             (1, 16, 32, 512, 512)])
def test_fp16_two_piece_forward_convolution_has_fp32_class_error(ops, monkeypatch, N, H, W, Cin, Cout):
    """Forward 3x3 / stride 1 convolutions run on TWO fp16 pieces per operand (three v_mfma_f32_32x32x16_f16 per product
    instead of six bf16 ones; conv_split.hip, split2h_quad).  Against a float64 convolution of the same fp32 inputs: error
    <= 1.5x the exact-fp32 MFMA kernel's (+ floor), no rounding bias, and it must really be another kernel than the three-
    piece bf16 one (the results differ in the last bits, both within the budget); with the BatchNorm + ReLU prologue and the
    statistics epilogue; small and large magnitudes inside the documented domain."""
    monkeypatch.setenv("DCS_KSPLIT", "0")
    monkeypatch.setenv("DCS_X3W_MIN", "1")                 # the weight-fragment kernel on these small maps
    ops.new_step(True)                                      # as a training-mode forward does (an eval forward keeps bf16)
    x = rnd(N, H, W, Cin, seed=141)
    x[0, :2] *= 300.0                                       # large activations
    x[-1, -2:] *= 1e-3                                      # and tiny ones
    w = cl(rnd(Cout, Cin, 3, 3, seed=142, scale=0.05))
    w[:3] *= 40.0
    w[3:6] *= 1e-3
    ref = E.conv_fwd(x.double(), w.double(), 1, 1)
    xd, wd = x.to(DEV), cl(w.to(DEV))
    g = ops.geom_fwd(N, H, W, Cin, Cout, 3, 3, 1, 1)
    assert ops.x3_ok(g) and ops.x3w_ok(g)
    got, st = ops.conv_fwd(xd, wd, 1, 1, want_stats=True)
    got2, st2 = ops.conv_fwd(xd, wd, 1, 1, want_stats=True)
    assert torch.equal(got, got2) and torch.equal(st, st2)
    monkeypatch.setenv("DCS_X2H", "0")
    b3 = ops.conv_fwd(xd, wd, 1, 1)
    monkeypatch.setenv("DCS_CONV_X3", "0")
    f32 = ops.conv_fwd(xd, wd, 1, 1)
    monkeypatch.delenv("DCS_CONV_X3"); monkeypatch.delenv("DCS_X2H")
    got, b3, f32 = got.cpu(), b3.cpu(), f32.cpu()
    assert not torch.equal(got, b3)
    scale = float(ref.abs().max())
    e_h, e_3, e_32 = (float((t.double() - ref).abs().max()) / scale for t in (got, b3, f32))
    l_h, l_32 = (float((t.double() - ref).norm() / ref.norm()) for t in (got, f32))
    print(f"max-rel fp16x2 {e_h:.3e} bf16x3 {e_3:.3e} fp32 {e_32:.3e} | l2-rel fp16x2 {l_h:.3e} fp32 {l_32:.3e}")
    assert e_h <= 1.5 * e_32 + 1e-7 and l_h <= 1.5 * l_32 + 1e-8
    err = (got.double() - ref)[:, 3:H - 3, :, 6:]           # the part of ordinary magnitude (equal weights in the bias measure)
    assert abs(float(err.sum() / err.abs().sum())) < 0.03, float(err.sum() / err.abs().sum())
    close(st[0, 0], got.reshape(-1, Cout).double().mean(0).float(), 1e-4, "statistics epilogue (mean)")
    # BatchNorm + ReLU prologue: bitwise the convolution of the materialised activation
    gam, bet = (rnd(Cin, seed=143) * 0.1 + 1).to(DEV), (rnd(Cin, seed=144) * 0.1).to(DEV)
    bn = ops.bn_finalize(ops.colsum(xd.reshape(-1, Cin), moments=True), gam, bet, torch.zeros(Cin, device=DEV),
                         torch.ones(Cin, device=DEV), N * H * W, True)
    assert torch.equal(ops.conv_fwd(xd, wd, 1, 1, pro=bn), ops.conv_fwd(ops.bn_act(xd, bn, relu=True), wd, 1, 1))


@pytest.mark.parametrize("N,H,W,C,mag", [(2, 16, 64, 64, 1e-6), (4, 16, 64, 128, 3e-9), (2, 8, 64, 256, 2e-3), (2, 16, 32, 128, 40.0)])
def test_fp16_two_piece_data_gradient_scales_by_the_tensor_maximum(ops, monkeypatch, N, H, W, C, mag):
    """Data gradients of 3x3 / stride 1 convolutions on two fp16 pieces: dy comes out of bn_bwd with the device word that
    holds max |dy| (integer atomicMax), the kernel scales it by the exact power of two that puts the maximum into
    [2^13, 2^14).  Gradient magnitudes from 1e-9 to 1e+1, heavy-tailed (a few entries 1000x the rest): error vs float64
    <= 1.5x the exact-fp32 MFMA kernel's, deterministic, and the maximum word is right."""
    monkeypatch.setenv("DCS_KSPLIT", "0")
    monkeypatch.setenv("DCS_X3W_MIN", "1")
    g = rnd(N, H, W, C, seed=151) * mag
    g.view(-1)[::997] *= 1000.0                               # heavy tail
    y = rnd(N, H, W, C, seed=152)
    gam, bet = (rnd(C, seed=153) * 0.1 + 1).to(DEV), (rnd(C, seed=154) * 0.1).to(DEV)
    yd, gd = y.to(DEV), g.to(DEV)
    bn = ops.bn_finalize(ops.colsum(yd.reshape(-1, C), moments=True), gam, bet, torch.zeros(C, device=DEV),
                         torch.ones(C, device=DEV), N * H * W, True)
    dy, _ = ops.bn_bwd(gd, yd, bn, gam, relu=True)
    assert hasattr(dy, "_dcs_max")
    assert float(dy._dcs_max.view(torch.float32)) == float(dy.abs().max())
    w = cl(rnd(C, C, 3, 3, seed=155, scale=0.05).to(DEV))
    wp = ops.pack_dgrad_weight(w)
    ref = E.conv_dgrad(dy.cpu().double(), E.pack_dgrad_weight(w.cpu().double()), (H, W), 1, 1)
    got = ops.conv_dgrad(dy, wp, (H, W), 1, 1)
    got2 = ops.conv_dgrad(dy, wp, (H, W), 1, 1)
    assert torch.equal(got, got2)
    monkeypatch.setenv("DCS_X2H", "0")
    b3 = ops.conv_dgrad(dy, wp, (H, W), 1, 1)
    monkeypatch.setenv("DCS_CONV_X3", "0")
    f32 = ops.conv_dgrad(dy, wp, (H, W), 1, 1)
    monkeypatch.delenv("DCS_CONV_X3"); monkeypatch.delenv("DCS_X2H")
    assert not torch.equal(got, b3)
    scale = float(ref.abs().max())
    e_h, e_3, e_32 = (float((t.cpu().double() - ref).abs().max()) / scale for t in (got, b3, f32))
    l_h, l_32 = (float((t.cpu().double() - ref).norm() / ref.norm()) for t in (got, f32))
    print(f"dgrad |dy|~{mag:g}: max-rel fp16x2 {e_h:.3e} bf16x3 {e_3:.3e} fp32 {e_32:.3e} | l2-rel fp16x2 {l_h:.3e} fp32 {l_32:.3e}")
    assert e_h <= 1.5 * e_32 + 1e-7 and l_h <= 1.5 * l_32 + 1e-8


@pytest.mark.parametrize("N,H,W,Cin,Cout,mag", [(4, 16, 64, 64, 64, 1e-6), (4, 16, 32, 128, 128, 3e-9), (8, 8, 32, 256, 128, 2e-3),
                                                 (2, 16, 48, 128, 256, 40.0)])
def test_fp16_two_piece_weight_gradient(ops, monkeypatch, N, H, W, Cin, Cout, mag):
    """Weight gradients of 3x3 / stride 1 convolutions on two fp16 pieces (rolling-window kernel; dy out of bn_bwd with its
    maximum word, the input with the BatchNorm + ReLU prologue): error vs a float64 weight gradient <= 1.5x the exact-fp32
    kernels', no rounding bias, deterministic, another kernel than the bf16 one."""
    ops.new_step(True)                                      # the input operand is a training-mode activation
    g = rnd(N, H, W, Cout, seed=161) * mag
    g.view(-1)[::997] *= 1000.0
    yb = rnd(N, H, W, Cout, seed=162)
    gam, bet = (rnd(Cout, seed=163) * 0.1 + 1).to(DEV), (rnd(Cout, seed=164) * 0.1).to(DEV)
    ybd = yb.to(DEV)
    bn_o = ops.bn_finalize(ops.colsum(ybd.reshape(-1, Cout), moments=True), gam, bet, torch.zeros(Cout, device=DEV),
                           torch.ones(Cout, device=DEV), N * H * W, True)
    dy, _ = ops.bn_bwd(g.to(DEV), ybd, bn_o, gam, relu=True)
    assert hasattr(dy, "_dcs_max")
    x = rnd(N, H, W, Cin, seed=165)
    xd = x.to(DEV)
    gi, bi_ = (rnd(Cin, seed=166) * 0.1 + 1).to(DEV), (rnd(Cin, seed=167) * 0.1).to(DEV)
    bn_i = ops.bn_finalize(ops.colsum(xd.reshape(-1, Cin), moments=True), gi, bi_, torch.zeros(Cin, device=DEV),
                           torch.ones(Cin, device=DEV), N * H * W, True)
    act = ops.bn_act(xd, bn_i, relu=True)
    ref = cl(torch.empty(Cout, Cin, 3, 3, dtype=torch.float64))
    E.conv_wgrad(act.cpu().double(), dy.cpu().double(), ref, 1, 1, False)
    got, got2, b3, f32 = (cl(torch.empty(Cout, Cin, 3, 3, device=DEV)) for _ in range(4))
    ops.conv_wgrad(xd, dy, got, 1, 1, False, pro=bn_i)
    ops.conv_wgrad(xd, dy, got2, 1, 1, False, pro=bn_i)
    monkeypatch.setenv("DCS_X2H", "0")
    ops.conv_wgrad(xd, dy, b3, 1, 1, False, pro=bn_i)
    monkeypatch.setenv("DCS_CONV_X3", "0")
    ops.conv_wgrad(xd, dy, f32, 1, 1, False, pro=bn_i)
    monkeypatch.delenv("DCS_CONV_X3"); monkeypatch.delenv("DCS_X2H")
    assert torch.equal(got, got2) and not torch.equal(got, b3)
    scale = float(ref.abs().max())
    err = got.cpu().double() - ref
    e_h, e_3, e_32 = float(err.abs().max()) / scale, float((b3.cpu().double() - ref).abs().max()) / scale, \
        float((f32.cpu().double() - ref).abs().max()) / scale
    bal = float(err.sum() / err.abs().sum())
    print(f"wgrad |dy|~{mag:g}: max-rel fp16x2 {e_h:.3e} bf16x3 {e_3:.3e} fp32 {e_32:.3e} sign balance {bal:+.3f}")
    assert e_h <= 1.5 * e_32 + 1e-7
    assert abs(bal) < 0.05, bal


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s,mag", [
    (4, 16, 24, 64, 128, 3, 2, 1e-6), (2, 20, 36, 256, 128, 1, 1, 2e-3), (2, 33, 29, 128, 128, 3, 1, 40.0),
    (2, 19, 21, 128, 80, 3, 1, 3e-9), (4, 16, 32, 64, 128, 1, 2, 1e-4),
])
def test_fp16_two_piece_weight_gradient_generic_kernel(ops, monkeypatch, N, H, W, Cin, Cout, k, s, mag):
    """Stride-2, 1x1 and ragged-width weight gradients (conv_wgrad_x3_kernel<BT, 2>: per-tap blocks) on two fp16 pieces, dy
    scaled by the maximum word: error vs float64 <= 1.5x the exact-fp32 kernel's, no rounding bias, deterministic."""
    ops.new_step(True)
    pad = k // 2
    OH, OW = ops.out_size(H, k, s, pad), ops.out_size(W, k, s, pad)
    dy = rnd(N, OH, OW, Cout, seed=171) * mag
    dy.view(-1)[::991] *= 300.0
    x = rnd(N, H, W, Cin, seed=172)
    dyd, xd = dy.to(DEV), x.to(DEV)
    ops.tag_max(dyd)
    ref = cl(torch.empty(Cout, Cin, k, k, dtype=torch.float64))
    E.conv_wgrad(x.double(), dy.double(), ref, s, pad, False)
    got, got2, b3, f32 = (cl(torch.empty(Cout, Cin, k, k, device=DEV)) for _ in range(4))
    ops.conv_wgrad(xd, dyd, got, s, pad, False)
    ops.conv_wgrad(xd, dyd, got2, s, pad, False)
    monkeypatch.setenv("DCS_X2H", "0")
    ops.conv_wgrad(xd, dyd, b3, s, pad, False)
    monkeypatch.setenv("DCS_CONV_X3", "0")
    ops.conv_wgrad(xd, dyd, f32, s, pad, False)
    monkeypatch.delenv("DCS_CONV_X3"); monkeypatch.delenv("DCS_X2H")
    assert torch.equal(got, got2) and not torch.equal(got, b3)
    scale = float(ref.abs().max())
    err = got.cpu().double() - ref
    e_h, e_3, e_32 = (float((t.cpu().double() - ref).abs().max()) / scale for t in (got, b3, f32))
    bal = float(err.sum() / err.abs().sum())
    e3 = b3.cpu().double() - ref
    bal3 = float(e3.sum() / e3.abs().sum())
    print(f"wgrad k{k} s{s} |dy|~{mag:g}: max-rel fp16x2 {e_h:.3e} bf16x3 {e_3:.3e} fp32 {e_32:.3e} sign balance {bal:+.3f} "
          f"(bf16x3 {bal3:+.3f})")
    assert e_h <= 1.5 * e_32 + 1e-7
    # uncancelled, the matrix core's downward rounding shows as -0.2 ... -0.5 here (tools/conv_bias_probe.py); a few hundred
    # pixels in two splits cancel it to a few percent
    assert abs(bal) < 0.08, bal


@pytest.mark.parametrize("N,H,W,mag", [(2, 64, 96, 1e-5), (3, 32, 64, 30.0)])
def test_fp16_two_piece_stem_weight_gradient(ops, monkeypatch, N, H, W, mag):
    """The 7x7 stem weight gradient (stem_wgrad_x3_kernel<2>) with dy out of the fused BatchNorm / pool backward and its
    maximum word: vs float64 <= 1.5x the exact-fp32 kernel's error, deterministic, another kernel than the bf16 one."""
    ops.new_step(True)
    img = rnd(N, H, W, 4, seed=181)
    img[..., 3] = 0
    w = rnd(64, 3, 7, 7, seed=182) * 0.1
    OH, OW = H // 2, W // 2
    dy = rnd(N, OH, OW, 64, seed=183) * mag
    dy.view(-1)[::977] *= 200.0
    pd, dyd = img.to(DEV), dy.to(DEV)
    ops.tag_max(dyd)
    dref = torch.zeros(64, 7, 8, 4, dtype=torch.float64)
    E.stem_wgrad(img.double(), dy.double(), dref, False)
    got, got2, b3, f32 = (torch.zeros(64, 7, 8, 4, device=DEV) for _ in range(4))
    ops.stem_wgrad(pd, dyd, got, False)
    ops.stem_wgrad(pd, dyd, got2, False)
    monkeypatch.setenv("DCS_X2H", "0")
    ops.stem_wgrad(pd, dyd, b3, False)
    monkeypatch.setenv("DCS_CONV_X3", "0")
    ops.stem_wgrad(pd, dyd, f32, False)
    monkeypatch.delenv("DCS_CONV_X3"); monkeypatch.delenv("DCS_X2H")
    assert torch.equal(got, got2) and not torch.equal(got, b3)
    wz = torch.zeros(64, 3, 7, 7)
    refw = E.unpack_stem_weight(dref, wz.double())
    scale = float(refw.abs().max())
    e_h, e_3, e_32 = (float((ops.unpack_stem_weight(t, cl(wz.to(DEV))).cpu().double() - refw).abs().max()) / scale
                      for t in (got, b3, f32))
    print(f"stem wgrad |dy|~{mag:g}: max-rel fp16x2 {e_h:.3e} bf16x3 {e_3:.3e} fp32 {e_32:.3e}")
    assert e_h <= 1.5 * e_32 + 1e-7


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,s", [
    (4, 24, 40, 64, 64, 3, 1), (2, 33, 29, 128, 128, 3, 1), (4, 16, 24, 64, 128, 3, 2), (2, 20, 36, 256, 128, 1, 1),
    (8, 12, 16, 512, 512, 3, 1), (2, 19, 21, 128, 80, 3, 1),
])
def test_split_bf16_weight_gradient_has_fp32_class_error(ops, monkeypatch, N, H, W, Cin, Cout, k, s):
    """dcs_conv_wgrad_x3 against the exact-fp32 kernels, both measured against a float64 weight gradient of the same
    inputs: error <= 1.5x the fp32 kernel's (+ floor), no rounding bias (signed / absolute error sum), deterministic."""
    pad = k // 2
    x = rnd(N, H, W, Cin, seed=111)
    OH, OW = E.conv_fwd(torch.zeros(1, H, W, Cin), cl(torch.zeros(Cout, Cin, k, k)), s, pad).shape[1:3]
    dy = rnd(N, OH, OW, Cout, seed=112)
    ref = cl(torch.empty(Cout, Cin, k, k, dtype=torch.float64))
    E.conv_wgrad(x.double(), dy.double(), ref, s, pad, False)
    xd, dyd = x.to(DEV), dy.to(DEV)
    got, got2, f32 = (cl(torch.empty(Cout, Cin, k, k, device=DEV)) for _ in range(3))
    ops.conv_wgrad(xd, dyd, got, s, pad, False)
    ops.conv_wgrad(xd, dyd, got2, s, pad, False)
    monkeypatch.setenv("DCS_CONV_X3", "0")
    ops.conv_wgrad(xd, dyd, f32, s, pad, False)
    monkeypatch.delenv("DCS_CONV_X3")
    assert torch.equal(got, got2) and not torch.equal(got, f32)
    scale = float(ref.abs().max())
    err = got.cpu().double() - ref
    e_x3, e_32 = float(err.abs().max()) / scale, float((f32.cpu().double() - ref).abs().max()) / scale
    bal = float(err.sum() / err.abs().sum())
    print(f"wgrad max-rel x3 {e_x3:.3e} fp32 {e_32:.3e} sign balance {bal:+.3f}")
    assert e_x3 <= 1.5 * e_32 + 1e-7
    assert abs(bal) < 0.05, bal


@pytest.mark.parametrize("N,H,W,Cin,Cout", [
    (2, 16, 64, 64, 64), (1, 8, 32, 64, 64), (3, 24, 32, 48, 64), (2, 16, 64, 64, 48),      # 8 x 32-pixel tiles, 64 channels wide
    (16, 16, 64, 128, 128), (8, 12, 96, 256, 128), (8, 8, 64, 128, 256),                    # 4 x 32-pixel tiles, 128 wide
])
def test_split_bf16_convolution_with_resident_halo(ops, monkeypatch, libopt, N, H, W, Cin, Cout):
    """conv3x3_x3_kernel (input halo resident in LDS, nine taps per staged chunk) forced on small maps (DCS_X3_HALO=2)
    against the per-tap split-bf16 kernel (DCS_X3_HALO=0) and float64: forward with statistics, BatchNorm + ReLU prologue
    (bitwise equal to the materialised activation), data gradient accumulating into a tensor with the BatchNorm-backward
    sums; deterministic; fp32-class error without bias."""
    monkeypatch.setenv("DCS_KSPLIT", "0")                 # the halo kernel has no K splits
    monkeypatch.setenv("DCS_X2H", "0")                    # the three-piece bf16 kernels are compared here
    x = rnd(N, H, W, Cin, seed=121)
    w = cl(rnd(Cout, Cin, 3, 3, seed=122, scale=0.05))
    ref = E.conv_fwd(x.double(), w.double(), 1, 1)
    xd, wd = x.to(DEV), cl(w.to(DEV))
    gam, bet = (rnd(Cin, seed=123) * 0.1 + 1).to(DEV), (rnd(Cin, seed=124) * 0.1).to(DEV)
    bn = ops.bn_finalize(ops.colsum(xd.reshape(-1, Cin), moments=True), gam, bet, torch.zeros(Cin, device=DEV),
                         torch.ones(Cin, device=DEV), N * H * W, True)
    dy = rnd(N, H, W, Cout, seed=125).to(DEV)
    base = rnd(N, H, W, Cin, seed=126).to(DEV)
    wp = ops.pack_dgrad_weight(wd)
    res = {}
    for mode in ("2", "0"):
        libopt("x3_halo", mode)
        y, st = ops.conv_fwd(xd, wd, 1, 1, want_stats=True)
        y2, st2 = ops.conv_fwd(xd, wd, 1, 1, want_stats=True)
        assert torch.equal(y, y2) and torch.equal(st, st2)
        yp = ops.conv_fwd(xd, wd, 1, 1, pro=bn)
        assert torch.equal(yp, ops.conv_fwd(ops.bn_act(xd, bn, relu=True), wd, 1, 1))
        d, sums = ops.conv_dgrad(dy, wp, (H, W), 1, 1, out=base.clone(), accumulate=True, bnb=(xd, None, bn, True))
        res[mode] = (y.cpu(), st.cpu(), d.cpu(), sums.cpu())
    (ya, sa, da, qa), (yb, sb, db, qb) = res["2"], res["0"]
    assert not torch.equal(ya, yb)                        # two different kernels really ran
    scale = float(ref.abs().max())
    ea, eb = ya.double() - ref, yb.double() - ref
    assert float(ea.abs().max()) / scale <= 1.5 * float(eb.abs().max()) / scale + 1e-7
    assert abs(float(ea.sum() / ea.abs().sum())) < 0.03
    close(sa, sb, 2e-5, "batch statistics"); close(da, db, 2e-5, "data gradient"); close(qa, qb, 2e-5, "BatchNorm-backward sums")


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(4, 16, 64, 128, 128), (2, 8, 32, 256, 128), (1, 12, 96, 128, 256), (2, 16, 64, 64, 64)])
def test_halo_kernel_with_weight_fragments_from_global_memory(ops, monkeypatch, libopt, N, H, W, Cin, Cout):
    """dcs_conv3x3_x3w (weight fragments straight from global memory in dcs_split_weight_frag layout, no per-tap barrier)
    performs exactly the arithmetic of conv3x3_x3_kernel: outputs, batch statistics, prologue and the data gradient with
    BatchNorm-backward sums must be BITWISE those of the LDS-staged halo kernel (DCS_X3_HALO=2 forces it on small maps)."""
    import ctypes as C
    from dcs_amd.ops import _p, _call, _stream
    monkeypatch.setenv("DCS_KSPLIT", "0")
    monkeypatch.setenv("DCS_X2H", "0")                    # the three-piece bf16 kernels are compared here
    libopt("x3_halo", 2)
    x = rnd(N, H, W, Cin, seed=131).to(DEV)
    w = cl(rnd(Cout, Cin, 3, 3, seed=132, scale=0.05).to(DEV))
    gam, bet = (rnd(Cin, seed=133) * 0.1 + 1).to(DEV), (rnd(Cin, seed=134) * 0.1).to(DEV)
    bn = ops.bn_finalize(ops.colsum(x.reshape(-1, Cin), moments=True), gam, bet, torch.zeros(Cin, device=DEV),
                         torch.ones(Cin, device=DEV), N * H * W, True)
    g = ops.geom_fwd(N, H, W, Cin, Cout, 3, 3, 1, 1)
    ref, st = ops.conv_fwd(x, w, 1, 1, want_stats=True, pro=bn)
    y = torch.empty_like(ref)
    part, G, G1 = ops._stats_buffer(N * H * W, Cout, x.device)
    _call("dcs_conv3x3_x3w", _p(x), _p(ops.split_weight_frag(ops.krsc(w))), None, _p(y), C.byref(g), 0, _p(part), _p(bn), None, None,
          None, 0, None, _stream())
    assert torch.equal(y, ref) and torch.equal(ops._stats_reduce(part, G, G1, Cout, N * H * W), st)
    # data gradient accumulating into a tensor, with the BatchNorm-backward sums
    dy = rnd(N, H, W, Cout, seed=135).to(DEV)
    base = rnd(N, H, W, Cin, seed=136).to(DEV)
    wp = ops.pack_dgrad_weight(w)
    d_ref, s_ref = ops.conv_dgrad(dy, wp, (H, W), 1, 1, out=base.clone(), accumulate=True, bnb=(x, None, bn, True))
    gd = ops.geoms_dgrad(N, H, W, Cin, Cout, 3, 3, 1, 1)[0]
    d = base.clone()
    tiles = -(-(N * H * W) // 128)
    partd = torch.empty((tiles, 2, Cin), device=DEV)
    _call("dcs_conv3x3_x3w", _p(dy), _p(ops.split_weight_frag(wp)), None, _p(d), C.byref(gd), 1, _p(partd), None, _p(x), None, _p(bn),
          1, None, _stream())
    sums = torch.empty((2, Cin), device=DEV)
    _call("dcs_colsum_final", _p(partd), _p(sums), 1, tiles, Cin, 1.0, 0.0, _stream())
    assert torch.equal(d, d_ref) and torch.equal(sums, s_ref)


LEVEL_MAPS = [
    # N, (H, W) per level, Cin, Cout, k, stride
    (2, [(32, 64), (16, 32), (8, 16)], 64, 64, 3, 1),       # halo kernel on the large level(s), per-tap on the small
    (2, [(16, 64), (8, 32), (4, 16)], 128, 128, 3, 1),
    (1, [(24, 40), (12, 20), (6, 10)], 64, 128, 3, 2),      # stride 2: four parity classes per level in the data gradient
    (2, [(12, 20), (6, 10), (3, 5)], 256, 128, 1, 1),       # bottleneck projection
    (1, [(6, 10), (3, 5), (2, 3)], 256, 512, 3, 2),         # K-split launches on the small levels
]


@pytest.mark.parametrize("N,maps,Cin,Cout,k,s", LEVEL_MAPS)
def test_level_batched_launches_are_bitwise_the_per_level_launches(ops, monkeypatch, N, maps, Cin, Cout, k, s):
    """ops.level_batch: the three pyramid levels of a layer through dcs_conv_gather_x3_multi / dcs_conv3x3_x3w_multi /
    dcs_conv_wgrad_x3_multi (one grid for the levels that select the same kernel) against the same operations launched
    level after level (DCS_LEVEL_BATCH=0): forward with BatchNorm statistics and prologue, data gradient with the
    BatchNorm-backward sums, weight gradient accumulated over the levels into one dW -- all BITWISE equal."""
    pad = k // 2
    w = cl(rnd(Cout, Cin, k, k, seed=1, scale=0.05)).to(DEV)
    xs = [rnd(N, h, wd, Cin, seed=10 + i).to(DEV) for i, (h, wd) in enumerate(maps)]
    pros = [torch.stack([1 + 0.1 * rnd(Cin, seed=20 + i), 0.1 * rnd(Cin, seed=30 + i), torch.zeros(Cin), torch.ones(Cin)]).to(DEV)
            for i in range(len(maps))]

    def run():
        out = []
        wp = ops.pack_dgrad_weight(w)
        dw = torch.empty_like(w)
        ys = [None] * len(xs)
        with ops.level_batch() as lb:
            for i, x in enumerate(xs):
                lb.level(i)
                y, st = ops.conv_fwd(x, w, s, pad, want_stats=True, pro=pros[i] if ops.pro_ok(Cin) else None)
                ys[i] = y
                out += [y, st]
            lb.flush()
            for i in reversed(range(len(xs))):
                lb.level(i)
                dy = ys[i] * 0.5 + 0.25
                bn = torch.stack([torch.ones(Cin), torch.zeros(Cin), torch.zeros(Cin), torch.ones(Cin)]).to(DEV)
                gx, sums = ops.conv_dgrad(dy, wp, xs[i].shape[1:3], s, pad, bnb=(xs[i], None, bn, True))
                ops.conv_wgrad(xs[i], dy, dw, s, pad, i != len(xs) - 1, pro=pros[i] if ops.pro_ok(Cin) else None)
                out += [gx] + ([sums] if sums is not None else [])
            lb.flush()
        out.append(dw)
        torch.cuda.synchronize()
        return [t.clone() for t in out]

    monkeypatch.setenv("DCS_LEVEL_BATCH", "0")
    before = dict(ops.launch_counts)
    ref = run()
    assert ops.launch_counts == before                      # nothing went through a batch
    monkeypatch.setenv("DCS_LEVEL_BATCH", "1")
    got = run()
    assert ops.launch_counts["multi"] > before["multi"], ops.launch_counts
    assert len(ref) == len(got)
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # and the per-level results themselves are right (forward of level 0 against the CPU statement)
    y0 = E.conv_fwd(xs[0].cpu(), w.cpu(), s, pad, pro=pros[0].cpu() if ops.pro_ok(Cin) else None)
    close(got[0], y0, 2e-5, "level 0 forward")


def test_level_batched_stem_is_bitwise_the_per_level_stem(ops, monkeypatch):
    """The 7x7 stem of the three pyramid levels and its weight gradient through the multi entries."""
    img = (rnd(2, 3, 128, 256, seed=3) * 40 + 100).to(DEV)
    w = cl(rnd(64, 3, 7, 7, seed=4, scale=0.05)).to(DEV)
    mean = torch.tensor([73.15, 82.90, 72.3], device=DEV)
    std = torch.tensor([47.67, 48.49, 47.73], device=DEV)

    def run():
        pyr = ops.normalize_pyramid(img, mean, std)
        wst = ops.pack_stem_weight(w)
        dwst = torch.empty((64, 7, 8, 4), device=DEV)
        out, ys = [], []
        with ops.level_batch() as lb:
            for i, p in enumerate(pyr):
                lb.level(i)
                y, st = ops.stem_conv(p, wst, want_stats=True)
                ys.append(y)
                out += [y, st]
            lb.flush()
            for i in reversed(range(3)):
                lb.level(i)
                ops.stem_wgrad(pyr[i], ys[i] * 0.5 + 0.1, dwst, i != 2)
            lb.flush()
        torch.cuda.synchronize()
        return [t.clone() for t in out + [dwst]]

    monkeypatch.setenv("DCS_LEVEL_BATCH", "0")
    ref = run()
    monkeypatch.setenv("DCS_LEVEL_BATCH", "1")
    before = ops.launch_counts["multi"]
    got = run()
    assert ops.launch_counts["multi"] >= before + 2
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


def test_streaming_access_paths_are_bitwise_the_default_paths(ops, monkeypatch, libopt):
    """Tensors of >= 256 MiB go through non-temporal loads / stores (dcs_common.h: dcs_streams) in the single-pass kernels
    and in the convolution epilogue.  The arithmetic is the same, only the cache policy differs: with the threshold forced
    to 0 (every tensor streams) each operation must return BITWISE what it returns with streaming off -- and the default
    path is what all other tests hold against the CPU statements."""
    N, H, W, C = 2, 24, 40, 64
    x = rnd(N, H, W, C, seed=1).to(DEV)
    g = rnd(N, H, W, C, seed=2).to(DEV)
    r = rnd(N, H, W, C, seed=3).to(DEV)
    w = cl(rnd(128, C, 3, 3, seed=4, scale=0.05)).to(DEV)
    bn = torch.stack([1 + 0.1 * rnd(C, seed=5), 0.1 * rnd(C, seed=6), 0.1 * rnd(C, seed=7), 1 + 0.1 * rnd(C, seed=8).abs()]).to(DEV)
    gamma = (1 + 0.1 * rnd(C, seed=9)).to(DEV)
    low = rnd(N, H // 2, W // 2, C, seed=10).to(DEV)

    def run():
        out = []
        y, st = ops.conv_fwd(x, w, 1, 1, want_stats=True, pro=bn)
        out += [y, st]
        wp = ops.pack_dgrad_weight(w)
        gx, sums = ops.conv_dgrad(y, wp, (H, W), 1, 1, bnb=(x, r, bn, False))
        out += [gx, sums]
        acc = g.clone()
        ops.conv_dgrad(y, wp, (H, W), 1, 1, out=acc, accumulate=True)
        out.append(acc)
        out.append(ops.bn_act(x, bn, r=r, bn2=bn, relu=True))
        dy, gm = ops.bn_bwd(g, x, bn, gamma, masksrc=r, want_gm=True)
        out += [dy, gm]
        out.append(ops.bn_bwd(g, x, bn, gamma, relu=True)[0])
        pooled, idx = ops.bn_relu_maxpool(x, bn)
        out += [pooled, idx]
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        out.append(ops.bn_pool_bwd(pooled * 0.5, idx, x, bn, gamma, dgamma=dg, dbeta=db))
        out += [dg, db]
        out.append(ops.upsample_add(low, [x, r], H, W))
        out.append(ops.colsum(x.reshape(-1, C), moments=True))
        a = x.clone()
        ops.axpy(a, g, 0.5)
        out.append(a)
        torch.cuda.synchronize()
        return [t.clone() for t in out if t is not None]

    monkeypatch.setenv("DCS_KSPLIT", "0")          # keep the fused BatchNorm-backward epilogue on this small map
    libopt("bn_nt", 0)
    ref = run()
    libopt("bn_nt", 1)
    libopt("nt_min_mb", 0)
    got = run()
    assert len(ref) == len(got) >= 17
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), i


@pytest.mark.parametrize("N,IH,IW,OH,OW,C,nsk", [(2, 8, 16, 16, 32, 128, 2), (3, 5, 7, 10, 14, 128, 1), (1, 16, 32, 32, 64, 64, 3),
                                                  (2, 3, 5, 6, 10, 256, 0)])
def test_upsample_add_with_batch_statistics(ops, monkeypatch, N, IH, IW, OH, OW, C, nsk):
    """dcs_upsample_add_stats: the same t as dcs_upsample_add (bitwise) and the batch statistics of t from the same
    launch, against the double-precision moments of t (the reduction pass it replaces is held to the same 2e-5)."""
    x = rnd(N, IH, IW, C, seed=1).to(DEV)
    sk = [rnd(N, OH, OW, C, seed=2 + i).to(DEV) for i in range(nsk)]
    t, st = ops.upsample_add(x, sk, OH, OW, want_stats=True)
    monkeypatch.setenv("DCS_UPSAMPLE_STATS", "0")
    t0, st0 = ops.upsample_add(x, sk, OH, OW, want_stats=True)
    assert torch.equal(t, t0)
    close(t, E.upsample_add(x.cpu(), [s.cpu() for s in sk], OH, OW), 2e-5, "t")
    v = t.double().reshape(-1, C)
    mom = torch.stack([v.mean(0), v.var(0, unbiased=False)])
    close(st.reshape(2, C), mom, 2e-5, "fused moments")
    close(st0.reshape(2, C), mom, 2e-5, "separate pass")


def test_conv_statistics_over_the_first_images(ops):
    """conv_fwd(stats_images=B): statistics of the first B images from a prefix of the epilogue's per-tile sums (the
    segmentation head's BatchNorm sees the first crop of a two-crop batch), one- and two-level reductions."""
    for (N, B, H, W, Cin, Cout) in [(4, 2, 16, 32, 64, 128), (6, 3, 256, 512, 64, 64)]:
        x = rnd(N, H, W, Cin, seed=5).to(DEV)
        w = cl(rnd(Cout, Cin, 3, 3, seed=6, scale=0.05)).to(DEV)
        y, st = ops.conv_fwd(x, w, 1, 1, want_stats=True, stats_images=B)
        v = y[:B].double().reshape(-1, Cout)
        close(st.reshape(2, Cout), torch.stack([v.mean(0), v.var(0, unbiased=False)]), 2e-5, (N, B, H, W))
        y_all, st_all = ops.conv_fwd(x, w, 1, 1, want_stats=True)
        assert torch.equal(y, y_all)
