"""GPU tests of the MFMA building blocks (csrc/selftest.hip, csrc/igemm.hip)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_mfma_fragment_layout():
    from bsed_amd import _lib as L
    rng = np.random.default_rng(0)
    K = 64
    A = rng.standard_normal((32, K)).astype(np.float32)
    B = rng.standard_normal((K, 32)).astype(np.float32)  # asymmetric on purpose
    a, b = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    c = torch.zeros((32, 32), device="cuda")
    L.call("bsed_selftest_mfma", L.ptr(a), L.ptr(b), L.ptr(c), L.c_int(K), L.stream())
    ref = A.astype(np.float64) @ B.astype(np.float64)
    np.testing.assert_allclose(c.cpu().numpy(), ref, atol=1e-4)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,W,cin,co", [(2, 13, 64, 16, 32), (1, 37, 32, 32, 64), (2, 21, 16, 64, 128),
                                          (1, 19, 8, 128, 128), (2, 70, 4, 128, 128), (1, 131, 2, 128, 128),
                                          (1, 9, 16, 32, 16)])
def test_conv3x3_forward_stats_and_dgrad(B, H, W, cin, co):
    from bsed_amd import ops
    rng = np.random.default_rng(B * 1000 + H)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((co, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(co).astype(np.float32))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    xg, wg, bg = _nhwc(x).cuda(), w.cuda(), b.cuda()
    wpk = ops.pack_weight(wg, 9, cin, co, 1, 9, cin * 9)
    y, stats = ops.igemm(xg, wpk, co, B, H, W, cin, taps=ops.TAPS3x3, bias=bg, epilogue=ops.EPI_STATS)
    got = y.cpu().double()
    np.testing.assert_allclose(got.numpy(), _nhwc(ref).numpy(), atol=2e-5)
    s = stats.double().sum(0).cpu()
    np.testing.assert_allclose(s[0].numpy(), ref.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[1].numpy(), (ref ** 2).sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    # data gradient = conv with flipped taps / swapped channel roles
    dy = torch.from_numpy(rng.standard_normal((B, co, H, W)).astype(np.float32))
    dref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1)
    wd = ops.pack_weight(wg, 9, co, cin, 1, cin * 9, 9)
    dx, _ = ops.igemm(_nhwc(dy).cuda(), wd, cin, B, H, W, co, taps=[(-a, -c) for a, c in ops.TAPS3x3])
    np.testing.assert_allclose(dx.cpu().double().numpy(), _nhwc(dref).numpy(), atol=5e-5)
    # weight gradient: exact fp32 matrix cores and split-fp32 (bf16x3) operands
    wref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1)
    for mode, tol in (("fp32", 1e-5), ("bf16x3", 4e-5)):
        part, G, KP, NP = ops.wgrad(xg, _nhwc(dy).cuda(), B, H, W, cin, co, taps=ops.TAPS3x3, mode=mode)
        dw = torch.zeros_like(wg)
        ops.reduce_partials(part, G, 9, KP, NP, cin, co, dw, 1, 9, cin * 9)
        err = float((dw.cpu().double() - wref).norm() / wref.norm())
        assert err < tol, (mode, err)


@pytest.mark.parametrize("M,K,N", [(300, 128, 768), (129, 256, 768), (1000, 768, 128), (77, 768, 256)])
def test_plain_gemm_and_tn_wgrad(M, K, N):
    from bsed_amd import ops
    rng = np.random.default_rng(M)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32))
    xg, wg, bg = x.cuda(), w.cuda(), b.cuda()
    wpk = ops.pack_weight(wg, 1, K, N, 0, 1, K)
    y, _ = ops.igemm(xg, wpk, N, 1, M, 1, K, bias=bg)
    ref = x.double() @ w.double().T + b.double()
    np.testing.assert_allclose(y.view(M, N).cpu().double().numpy(), ref.numpy(), atol=5e-5)
    if K > 256:
        return  # weight gradients only ever contract over <= 256 input features on this path
    dy = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
    wref = dy.double().T @ x.double()
    for mode, tol in (("fp32", 1e-5), ("bf16x3", 4e-5)):
        part, G, KP, NP = ops.wgrad(xg, dy.cuda(), 1, M, 1, K, N, mode=mode)
        dw = torch.zeros_like(wg)
        ops.reduce_partials(part, G, 1, KP, NP, K, N, dw, 0, 1, K)
        err = float((dw.cpu().double() - wref).norm() / wref.norm())
        assert err < tol, (mode, err)


def test_shifted_tap_wgrad_matches_gru_hidden_gradient_form():
    """dW_hh = sum_t dgh_t^T h_{t-1}: a 1-tap wgrad with a (-1, 0) / (+1, 0) offset and pitched operands"""
    from bsed_amd import ops
    rng = np.random.default_rng(3)
    B, T = 3, 150
    out = torch.from_numpy(rng.standard_normal((B, T, 256)).astype(np.float32))
    dgh = torch.from_numpy(rng.standard_normal((B, T, 768)).astype(np.float32))
    og, dg = out.cuda(), dgh.cuda()
    for dr, mode in ((0, "fp32"), (1, "fp32"), (0, "bf16x3"), (1, "bf16x3")):
        part, G, KP, NP = ops.wgrad(og, dg, B, T, 1, 128, 384, taps=((-1 if dr == 0 else 1, 0),), in_pitch=256,
                                    dy_pitch=768, in_offset=dr * 128, dy_offset=dr * 384, mode=mode)
        dw = torch.zeros((384, 128), device="cuda")
        ops.reduce_partials(part, G, 1, KP, NP, 128, 384, dw, 0, 1, 128)
        h = out[:, :, dr * 128:(dr + 1) * 128].double()
        hp = torch.zeros_like(h)
        if dr == 0:
            hp[:, 1:] = h[:, :-1]
        else:
            hp[:, :-1] = h[:, 1:]
        ref = torch.einsum("btj,btk->jk", dgh[:, :, dr * 384:(dr + 1) * 384].double(), hp)
        err = float((dw.cpu().double() - ref).norm() / ref.norm())
        assert err < (1e-5 if mode == "fp32" else 4e-5), (dr, mode, err)


def test_gru_forward_backward_vs_torch():
    from bsed_amd import ops
    torch.manual_seed(0)
    B, T = 3, 40
    gru = torch.nn.GRU(128, 128, bidirectional=True, batch_first=True)
    x = torch.randn(B, T, 128)
    x.requires_grad_()
    ref, _ = gru(x)
    dout = torch.randn(B, T, 256)
    ref.backward(dout)
    sd = gru.state_dict()
    w_ih = torch.cat([sd["weight_ih_l0"], sd["weight_ih_l0_reverse"]]).cuda()
    w_hh = torch.cat([sd["weight_hh_l0"], sd["weight_hh_l0_reverse"]]).contiguous().cuda()
    b_ih = torch.cat([sd["bias_ih_l0"], sd["bias_ih_l0_reverse"]]).cuda()
    b_hh = torch.cat([sd["bias_hh_l0"], sd["bias_hh_l0_reverse"]]).cuda()
    xg = x.detach().cuda()
    wpk = ops.pack_weight(w_ih, 1, 128, 768, 0, 1, 128)
    xp, _ = ops.igemm(xg, wpk, 768, 1, B * T, 1, 128, bias=b_ih)
    out, gates = ops.gru_fwd(xp.view(B, T, 768), w_hh, b_hh, B, T, save_gates=True)
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().numpy(), atol=2e-6)
    dxp, dgh, _, _ = ops.gru_bwd(dout.cuda(), out, gates, w_hh, B, T)
    wd = ops.pack_weight(w_ih, 1, 768, 128, 0, 128, 1)
    dx, _ = ops.igemm(dxp, wd, 128, 1, B * T, 1, 768)
    np.testing.assert_allclose(dx.view(B, T, 128).cpu().numpy(), x.grad.numpy(), atol=2e-5)
    db = torch.zeros(768, device="cuda")
    ops.colsum(dgh, B * T, 768, 768, db)
    ref_db = torch.cat([gru.bias_hh_l0.grad, gru.bias_hh_l0_reverse.grad])
    np.testing.assert_allclose(db.cpu().numpy(), ref_db.numpy(), atol=2e-5)


def test_mfma_bf16x3_fragment_layout_and_accuracy():
    from bsed_amd import _lib as L
    rng = np.random.default_rng(1)
    K = 128
    A = rng.standard_normal((32, K)).astype(np.float32)
    B = rng.standard_normal((K, 32)).astype(np.float32)
    a, b = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    c = torch.zeros((32, 32), device="cuda")
    L.call("bsed_selftest_mfma_bf16x3", L.ptr(a), L.ptr(b), L.ptr(c), L.c_int(K), L.stream())
    ref = A.astype(np.float64) @ B.astype(np.float64)
    err = np.abs(c.cpu().numpy() - ref).max()
    assert err < 5e-4, err       # ~2e-5 relative on terms of size ~1, sum of 128
    assert err / np.abs(ref).max() < 3e-5


@pytest.mark.parametrize("B,H,W,cin,co", [(2, 21, 16, 64, 128), (1, 19, 8, 128, 128), (2, 70, 4, 128, 128),
                                          (1, 37, 32, 32, 64), (1, 9, 16, 32, 16)])
def test_conv3x3_bf16x3_matches_fp64_reference(B, H, W, cin, co):
    """split-fp32 operands on the bf16 matrix cores: forward (+BN sums) and data gradient within 3e-5 of max|ref|"""
    from bsed_amd import ops
    rng = np.random.default_rng(B * 1000 + H)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((co, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(co).astype(np.float32))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    xg, wg, bg = _nhwc(x).cuda(), w.cuda(), b.cuda()
    w3 = ops.pack_weight3(wg, 9, cin, co, 1, 9, cin * 9)
    y, stats = ops.igemm3(xg, w3, co, B, H, W, cin, ops.TAPS3x3, bias=bg, epilogue=ops.EPI_STATS)
    err = float((y.cpu().double() - _nhwc(ref)).abs().max())
    assert err < 3e-5 * float(ref.abs().max()), err
    s = stats.double().sum(0).cpu()
    np.testing.assert_allclose(s[0].numpy(), ref.sum((0, 2, 3)).numpy(), rtol=1e-3, atol=2e-2)
    if co % 32 == 0:
        dy = torch.from_numpy(rng.standard_normal((B, co, H, W)).astype(np.float32))
        dref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1)
        wd3 = ops.pack_weight3(wg, 9, co, cin, 1, cin * 9, 9)
        dx, _ = ops.igemm3(_nhwc(dy).cuda(), wd3, cin, B, H, W, co, [(-a, -c) for a, c in ops.TAPS3x3])
        err = float((dx.cpu().double() - _nhwc(dref)).abs().max())
        assert err < 3e-5 * float(dref.abs().max()), err


@pytest.mark.gpu
@pytest.mark.parametrize("C,B,H,W,pool", [(32, 2, 24, 64, (2, 2)), (32, 1, 13, 16, (2, 2)), (32, 3, 27, 32, (1, 2)),
                                          (32, 2, 627, 64, (2, 2)), (32, 1, 21, 64, (2, 2)), (64, 2, 313, 32, (1, 2)),
                                          (64, 2, 27, 32, (1, 2)), (64, 1, 16, 16, (2, 2)), (64, 2, 9, 8, (1, 2))])
def test_glu_backward_split_fp32_matches_fp32_fused_kernel(C, B, H, W, pool):
    """csrc/glu3.hip (bf16 cores, split-fp32 operands, register-layout operands) vs csrc/glu_bwd.hip (fp32 cores) on
    the same inputs, dropout on, partial tiles (H % TH != 0) and odd pooled extents included"""
    from bsed_amd import ops
    g = torch.Generator().manual_seed(C + H)
    y = torch.randn(B, H, W, C, generator=g).cuda()
    scale = (torch.rand(C, generator=g) + 0.5).cuda()
    shift = (torch.randn(C, generator=g) * 0.3).cuda()
    w = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    bias = (torch.randn(C, generator=g) * 0.1).cuda()
    ph, pw = pool
    dpool = torch.randn(B, H // ph, W // pw, C, generator=g).cuda()
    wfwd = ops.pack_weight(w, 1, C, C, 0, 1, C)
    ref = ops.glu_bwd_fused(y, scale, shift, wfwd, w, bias, dpool, B, H, W, C, pool, 0.5, 103, 9)
    got = ops.glu_bwd3(y, scale, shift, w, bias, dpool, B, H, W, C, pool, 0.5, 103, 9)

    def finish(r):
        gq, pdw, pdb, st, G, slabs = r
        dw = torch.zeros(C, C, device="cuda")
        ops.reduce_partials(pdw, G * slabs, 1, C, C, C, C, dw, 0, C, 1)
        return gq, dw, pdb.sum(0)[0], st.sum(0)

    for name, a, b in zip(("g", "dW", "db", "bn sums"), finish(got), finish(ref)):
        err = float((a - b).norm() / b.norm())
        assert err < 3e-5, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("C,B,H,W,pool", [(32, 2, 24, 64, (2, 2)), (32, 1, 13, 16, (2, 2)), (32, 2, 627, 64, (2, 2)),
                                          (64, 2, 27, 32, (1, 2)), (64, 1, 17, 8, (2, 2)), (128, 2, 21, 16, (1, 2)),
                                          (128, 3, 9, 2, (1, 2)), (128, 1, 40, 4, (1, 2)), (64, 2, 10, 2, (2, 2)),
                                          (128, 1, 12, 8, (1, 1)), (128, 2, 313, 1, (2, 1)), (128, 3, 156, 1, (2, 1))])
def test_glu_forward_split_fp32_matches_torch(C, B, H, W, pool):
    """csrc/glu3.hip forward vs torch (dropout off) and vs the fp32-core kernel (dropout on: same masks)"""
    from bsed_amd import ops
    g = torch.Generator().manual_seed(C + H + W)
    y = torch.randn(B, H, W, C, generator=g).cuda()
    scale = (torch.rand(C, generator=g) + 0.5).cuda()
    shift = (torch.randn(C, generator=g) * 0.3).cuda()
    w = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    bias = (torch.randn(C, generator=g) * 0.1).cuda()
    assert ops.glu_fwd3_supported(W, C, pool)
    got = ops.glu_fwd3(y, scale, shift, w, bias, B, H, W, C, pool, 0.0, 101, 5)
    xn = y * scale + shift
    res = (xn @ w.t() + bias) * torch.sigmoid(xn)
    ref = torch.nn.functional.avg_pool2d(res.permute(0, 3, 1, 2), pool).permute(0, 2, 3, 1)
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    wpk = ops.pack_weight(w, 1, C, C, 0, 1, C)
    ref_d, _ = ops.igemm(y, wpk, C, B, H, W, C, bias=bias, epilogue=ops.EPI_GLU_POOL, a_scale=scale, a_shift=shift,
                         e_src=y, e_scale=scale, e_shift=shift, pool=pool, drop_p=0.5, rng_stream=101, seed=5)
    got_d = ops.glu_fwd3(y, scale, shift, w, bias, B, H, W, C, pool, 0.5, 101, 5)
    assert float((got_d - ref_d).abs().max()) < 4e-5 * max(1.0, float(ref_d.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,pool", [(2, 21, 16, (1, 2)), (3, 9, 2, (1, 2)), (1, 313, 8, (1, 2)), (2, 16, 4, (2, 2))])
def test_glu_backward_c128_split_fp32_matches_fp32_fused_kernel(B, H, W, pool):
    """bsed_glu_bwd3n (g, d_lin, db, BatchNorm sums) + 1-tap wgrad3 for dW vs csrc/glu_bwd.hip, dropout on"""
    from bsed_amd import ops
    C = 128
    g = torch.Generator().manual_seed(H + W)
    y = torch.randn(B, H, W, C, generator=g).cuda()
    scale = (torch.rand(C, generator=g) + 0.5).cuda()
    shift = (torch.randn(C, generator=g) * 0.3).cuda()
    w = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    bias = (torch.randn(C, generator=g) * 0.1).cuda()
    ph, pw = pool
    dpool = torch.randn(B, H // ph, W // pw, C, generator=g).cuda()
    wfwd = ops.pack_weight(w, 1, C, C, 0, 1, C)
    gr, pdw, pdb, st, G, slabs = ops.glu_bwd_fused(y, scale, shift, wfwd, w, bias, dpool, B, H, W, C, pool, 0.5, 104, 9)
    dw_ref = torch.zeros(C, C, device="cuda")
    ops.reduce_partials(pdw, G * slabs, 1, C, C, C, C, dw_ref, 0, C, 1)
    gq, dlin, pdb2, st2, G2 = ops.glu_bwd3n(y, scale, shift, w, bias, dpool, B, H, W, C, pool, 0.5, 104, 9)
    part, Gw, KP, NP = ops.wgrad(y, dlin, B, H, W, C, C, a_scale=scale, a_shift=shift, mode="bf16x3")
    dw = torch.zeros(C, C, device="cuda")
    ops.reduce_partials(part, Gw, 1, KP, NP, C, C, dw, 0, 1, C)
    for name, a, b in (("g", gq, gr), ("dW", dw, dw_ref), ("db", pdb2.sum(0)[0], pdb.sum(0)[0]),
                       ("bn sums", st2.sum(0), st.sum(0))):
        err = float((a - b).norm() / b.norm())
        assert err < 3e-5, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,N", [(2, 40, 64, 32), (3, 21, 16, 32), (1, 9, 8, 64), (2, 300, 64, 32)])
def test_conv_cin16_split_fp32_matches_torch(B, H, W, N):
    """bsed_igemm3s (all taps' weights resident in LDS, persistent workgroups) vs F.conv2d, with the BatchNorm sums"""
    from bsed_amd import ops
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, H, W, 16, generator=g).cuda()
    w = (torch.randn(N, 16, 3, 3, generator=g) * 0.1).cuda()
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    wt = ops.pack_weight3s(w, 9, N, 1, 9, 16 * 9)
    out, stats = ops.igemm3s(x, wt, N, B, H, W, ops.TAPS3x3, bias=bias, epilogue=ops.EPI_STATS)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    assert float((out.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    tot = stats.double().sum(0)
    assert float((tot[0] - ref.sum((0, 1, 2))).abs().max()) < 1e-3 * max(1.0, float(ref.sum((0, 1, 2)).abs().max()))
    assert float((tot[1] - (ref * ref).sum((0, 1, 2))).abs().max()) < 1e-4 * float((ref * ref).sum((0, 1, 2)).max())
    out2, _ = ops.igemm3s(x, wt, N, B, H, W, ops.TAPS3x3, bias=bias)
    assert torch.equal(out, out2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,N", [(2, 40, 64, 16), (3, 21, 16, 16), (2, 300, 64, 32)])
def test_dgrad_cin32_split_fp32_matches_torch(B, H, W, N):
    """igemm3s with two K steps per tap: data gradient of a 32-channel 3x3 convolution (flipped taps, transposed weight)"""
    from bsed_amd import ops
    g = torch.Generator().manual_seed(H + W + N)
    dy = torch.randn(B, H, W, 32, generator=g).cuda()
    w = (torch.randn(32, N, 3, 3, generator=g) * 0.1).cuda()        # conv weight (Cout = 32, Cin = N)
    wt = ops.pack_weight3s(w, 9, N, 1, N * 9, 9, K=32)              # [tap][k = cout][n = cin]
    flipped = [(-a, -b) for a, b in ops.TAPS3x3]
    assert ops.igemm3s_supported(W, 32)
    dx, _ = ops.igemm3s(dy, wt, N, B, H, W, flipped)
    ref = torch.nn.functional.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    assert float((dx.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,cin,co,affine", [
    (2, 45, 64, 16, 32, False),     # conv1: two taps share one MFMA tile (pack2)
    (2, 37, 32, 32, 64, True),      # 32-channel chunks, producer/consumer kernel, 5 slots
    (3, 50, 16, 64, 128, True),     # 9 slots, producer/consumer kernel, 16-wide tiles
    (2, 61, 8, 128, 128, False),    # 9 slots, two channel chunks (grid.z = 2)
    (2, 70, 4, 128, 128, True),     # tall 34 x 6 patch
    (2, 130, 2, 128, 128, False),   # 66 x 4 patch: 32-channel chunks
    (1, 19, 16, 64, 64, False),     # one ragged tile row only
])
def test_conv3x3_weight_gradient_bf16x3_matches_fp64_reference(B, H, W, cin, co, affine, monkeypatch):
    """dW of a 3x3 convolution from the split-fp32 weight-gradient kernels (transposing LDS reads; the
    producer/consumer form where it applies and the single-buffer form forced through BSED_WGRAD3_NOPIPE), with and
    without the fused per-channel affine on the staged activations (BatchNorm apply of the previous layer)."""
    from bsed_amd import ops
    rng = np.random.default_rng(B * 100 + H + W)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    dy = torch.from_numpy(rng.standard_normal((B, co, H, W)).astype(np.float32))
    sc = torch.from_numpy(rng.uniform(0.5, 1.5, cin).astype(np.float32))
    sh = torch.from_numpy(rng.standard_normal(cin).astype(np.float32))
    xin = x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) if affine else x.double()
    ref = torch.nn.grad.conv2d_weight(xin, (co, cin, 3, 3), dy.double(), padding=1)
    xg, dyg = _nhwc(x).cuda(), _nhwc(dy).cuda()
    kw = dict(a_scale=sc.cuda(), a_shift=sh.cuda()) if affine else {}
    got = {}
    for nopipe in ("", "1"):
        if nopipe:
            monkeypatch.setenv("BSED_WGRAD3_NOPIPE", "1")
        else:
            monkeypatch.delenv("BSED_WGRAD3_NOPIPE", raising=False)
        part, G, KP, NP = ops.wgrad(xg, dyg, B, H, W, cin, co, taps=ops.TAPS3x3, mode="bf16x3", **kw)
        dw = torch.zeros((co, cin, 3, 3), device="cuda")
        ops.reduce_partials(part, G, 9, KP, NP, cin, co, dw, 1, 9, cin * 9)
        got[nopipe] = dw.cpu()
        err = float((dw.cpu().double() - ref).norm() / ref.norm())
        assert err < 4e-5, (nopipe, err)
    # both forms build the same products in the same order; they differ only in how many partial slabs are summed
    np.testing.assert_allclose(got[""].numpy(), got["1"].numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,cin,co", [
    (2, 37, 16, 64, 128),    # every workgroup stores d_y (one input-channel chunk), ragged last tile row
    (2, 61, 8, 128, 128),    # two input-channel chunks: only chunk 0 stores d_y
    (3, 50, 32, 32, 64),     # 32-channel chunks, 5 slots
    (2, 37, 16, 48, 48),     # channel quads beyond CIN / N: clamped loads, zeroed LDS image, duplicate d_y stores
    (1, 70, 4, 128, 96),     # tall patch, N = 3 x 32
])
@pytest.mark.parametrize("act", ["f32", "bf16"])
def test_conv3x3_weight_gradient_with_batchnorm_backward_on_load(B, H, W, cin, co, act):
    """wgrad3p's producers form d_y = A g + B (y - mean) + C on load (BatchNorm backward, src/models/CNN.py:46-67's
    nn.BatchNorm2d under autograd) and write it out once for the data-gradient convolution.  The producers are software
    pipelined over tiles with clamped, branch-free loads: the cases cover ragged tile rows, one / two input-channel
    chunks and channel counts that leave idle channel quads; fp32 and bf16 activations."""
    from bsed_amd import ops
    rng = np.random.default_rng(7 * H + W + cin)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((B, co, H, W)).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal((B, co, H, W)).astype(np.float32))
    coef = torch.from_numpy(rng.uniform(0.5, 1.5, (3, co)).astype(np.float32))
    coef[1] *= 0.1
    coef[2] *= 0.01
    mean = torch.from_numpy(rng.standard_normal(co).astype(np.float32) * 0.2)
    dt = torch.bfloat16 if act == "bf16" else torch.float32
    xg, gg, yg = (_nhwc(t).to(dt).cuda() for t in (x, g, y))
    xq, gq, yq = (t.float().cpu().permute(0, 3, 1, 2).double() for t in (xg, gg, yg))   # what the kernel really reads
    A, Bc, Cc = (coef[i].double().view(1, -1, 1, 1) for i in range(3))
    dy_ref = A * gq + Bc * (yq - mean.double().view(1, -1, 1, 1)) + Cc
    dy_out = torch.full_like(gg, float("nan"))
    part, G, KP, NP = ops.wgrad(xg, gg, B, H, W, cin, co, taps=ops.TAPS3x3, mode="bf16" if act == "bf16" else "bf16x3",
                                bn_y=yg, bn_coef=coef.cuda(), bn_mean=mean.cuda(), dy_out=dy_out)
    dw = torch.zeros((co, cin, 3, 3), device="cuda")
    ops.reduce_partials(part, G, 9, KP, NP, cin, co, dw, 1, 9, cin * 9)
    torch.cuda.synchronize()
    tol_dy = 8e-3 if act == "bf16" else 2e-6      # bf16: d_y is rounded to bf16 (2^-9 relative) when stored
    err_dy = float((dy_out.float().cpu().permute(0, 3, 1, 2).double() - dy_ref).abs().max() / dy_ref.abs().max())
    assert err_dy < tol_dy, err_dy                 # every element written (no NaN left), none clobbered
    # the contraction uses the d_y it formed: fp32 values (split-fp32 products) or their bf16 roundings (bf16 mode)
    dy_used = dy_out.float().cpu().permute(0, 3, 1, 2).double() if act == "bf16" else dy_ref
    ref = torch.nn.grad.conv2d_weight(xq, (co, cin, 3, 3), dy_used, padding=1)
    err = float((dw.cpu().double() - ref).norm() / ref.norm())
    assert err < (3e-3 if act == "bf16" else 4e-5), err
