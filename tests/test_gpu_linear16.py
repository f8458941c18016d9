"""k_linear16 (one dense layer, fp32-accurate hi/lo-split products on the fp16 matrix cores) against a float64 reference of
the same nn.Linear (+ ReLU): the tolerance is that of a plain fp32 GEMM (relative 2e-6 of the row's |x| . |w| bound)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x, w, b, relu):
    y = x.double() @ w.double().t() + (b.double() if b is not None else 0)
    return torch.relu(y) if relu else y


def _check(y, x, w, b, relu, tol=3e-6):
    ref = _ref(x, w, b, relu)
    bound = x.double().abs() @ w.double().abs().t() + 1e-30          # magnitude of the accumulated products
    err = ((y.double() - ref).abs() / bound).max().item()
    assert err < tol, err


@pytest.mark.parametrize("M,K,N,relu", [(1000, 448, 448, True), (129, 432, 448, True), (5000, 448, 225, False), (77, 64, 16, False),
                                         (300, 256, 256, True), (4096, 448, 1, False), (257, 36, 40, True)])
def test_linear16_matches_float64(M, K, N, relu):
    from core import hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    packed, shape = ops.linear16_pack(w)
    y = ops.linear16(x, packed, shape, b, relu=relu)
    assert y.shape == (M, N)
    _check(y, x, w, b, relu)


def test_linear16_skip_layer_two_inputs_strided_views_and_device_count():
    from core import hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(5)
    M, K1, K2, N = 3000, 432, 448, 448
    xa, xb = torch.randn(M, K1, generator=g).to(DEV), torch.randn(M, K2, generator=g).to(DEV)
    w = (torch.randn(N, K1 + K2, generator=g) / 30).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    packed, shape = ops.linear16_pack(w, K1=K1)
    y = ops.linear16(xa, packed, shape, b, relu=True, x2=xb)
    _check(y, torch.cat([xa, xb], 1), w, b, True)
    # output into a column slice of a wider buffer; the neighbouring columns stay untouched
    wide = torch.full((M, 456), 7.0, device=DEV)
    ops.linear16(xa, packed, shape, b, relu=True, x2=xb, out=wide[:, :448])
    assert torch.equal(wide[:, :448], y) and bool((wide[:, 448:] == 7.0).all())
    # inputs that are column slices of wider buffers
    big = torch.randn(M, 900, generator=g).to(DEV)
    y2 = ops.linear16(big[:, :432], packed, shape, b, relu=True, x2=big[:, 440:888])
    _check(y2, torch.cat([big[:, :432], big[:, 440:888]], 1), w, b, True)
    # device-side row count (compacted lists): rows past the count are not written
    cnt = torch.tensor([1234], dtype=torch.int32, device=DEV)
    out = torch.full((M, N), -1.0, device=DEV)
    ops.linear16(xa, packed, shape, b, relu=True, x2=xb, out=out, count=cnt)
    assert torch.equal(out[:1234], y[:1234]) and bool((out[1234:] == -1.0).all())


def test_linear16_transposed_weight_is_the_backward_data_gemm():
    """dX = dY W: the same kernel with the weight read through its transpose"""
    from core import hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(9)
    M, K, N = 700, 256, 451
    dy = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(K, N, generator=g) / 16).to(DEV)          # nn.Linear(N -> K).weight
    packed, shape = ops.linear16_pack(w, transposed=True)
    assert shape == (N, K, 0)
    buf = torch.empty(M, 452, device=DEV)
    dx = ops.linear16(dy, packed, shape, None, out=buf[:, :N])
    _check(dx, dy, w.t().contiguous(), None, False)


def test_linear16_unaligned_rows_take_the_scalar_path():
    from core import hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(11)
    M, K, N = 500, 195, 256                                     # DANBO's first layer: 195 inputs, rows not 16-byte aligned
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / 14).to(DEV)
    packed, shape = ops.linear16_pack(w)
    _check(ops.linear16(x, packed, shape, None, relu=True), x, w, None, True)


def test_linear16_large_is_deterministic_and_rejects_bad_shapes():
    from core import hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(13)
    M = 1 << 17
    x = torch.randn(M, 448, generator=g).to(DEV)
    w = (torch.randn(448, 448, generator=g) / 21).to(DEV)
    packed, shape = ops.linear16_pack(w)
    a, b = ops.linear16(x, packed, shape, None, relu=True), ops.linear16(x, packed, shape, None, relu=True)
    assert torch.equal(a, b)
    _check(a[::97], x[::97], w, None, True)
    with pytest.raises(ValueError):
        ops.linear16_pack(torch.zeros(600, 64, device=DEV))
    with pytest.raises(ValueError):
        ops.linear16(x[:, :100], packed, shape)
    with pytest.raises(RuntimeError):
        ops.linear16(x.cpu(), packed, shape)


def test_linear16_random_layer_shapes():
    """rows 1 .. 70 000, inputs 4 .. 708, outputs 1 .. 512: every tile-pair count, both ring widths, ragged last tiles"""
    from core import hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(123)
    for trial in range(40):
        M = int(torch.randint(1, 70000, (1,), generator=g))
        K = int(torch.randint(1, 60, (1,), generator=g)) * 4 * (1 + trial % 3)
        N = int(torch.randint(1, 513, (1,), generator=g))
        x = torch.randn(M, K, generator=g).to(DEV)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        packed, shape = ops.linear16_pack(w)
        y = ops.linear16(x, packed, shape, b, relu=bool(trial & 1))
        # 2^-22 per product plus the fp16-subnormal floor of the lo parts (k_linear16.hip): visible only for K of a few columns
        _check(y, x, w, b, bool(trial & 1), tol=3e-6 if K >= 32 else 2e-5)


def test_fragment_order_buffer_round_trip():
    """FragBuffer: element (16 g + n, 32 s + 16 h + 4 q + i) at [g][s][h][q][n][i] (include/danbo_hip.h)"""
    from core import hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(300, 96, generator=g).to(DEV)
    fb = ops.FragBuffer.from_rows(x)
    assert fb.data.numel() == 384 * 96 and torch.equal(fb.rows(), x)
    d = fb.data.view(-1, 3, 2, 4, 16, 4)
    assert float(d[2, 1, 1, 3, 5, 2]) == float(x[16 * 2 + 5, 32 * 1 + 16 * 1 + 4 * 3 + 2])


@pytest.mark.parametrize("W,M", [(448, 40000), (256, 1000), (64, 129)])
def test_linear16_trunk_in_fragment_order_matches_the_row_major_chain(W, M):
    """rows -> fragments, fragments -> fragments, [rows | fragments] -> fragments (skip layer), fragments -> rows: the chain of
    layers the A-NeRF engine runs, against the same chain on row-major buffers (each layer sums the same products; only the
    position of an input inside its 32-wide k-step differs) and against float64"""
    from core import hip_ops as ops
    g = torch.Generator(device="cpu").manual_seed(W + M)
    K0, Nh = 432, 225 if W >= 225 else 64
    x0 = torch.randn(M, K0, generator=g).to(DEV)
    ws = [(torch.randn(W, K0, generator=g) / K0 ** 0.5).to(DEV), (torch.randn(W, W, generator=g) / W ** 0.5).to(DEV),
          (torch.randn(W, K0 + W, generator=g) / (K0 + W) ** 0.5).to(DEV), (torch.randn(Nh, W, generator=g) / W ** 0.5).to(DEV)]
    bs = [torch.randn(w.shape[0], generator=g).to(DEV) * 0.1 for w in ws]
    # row-major chain
    pk = [ops.linear16_pack(ws[0]), ops.linear16_pack(ws[1]), ops.linear16_pack(ws[2], K1=K0), ops.linear16_pack(ws[3])]
    h1 = ops.linear16(x0, *pk[0], bs[0], relu=True)
    h2 = ops.linear16(h1, *pk[1], bs[1], relu=True)
    h3 = ops.linear16(x0, *pk[2], bs[2], relu=True, x2=h2)
    yr = ops.linear16(h3, *pk[3], bs[3])
    # fragment-order chain
    pf = [ops.linear16_pack(ws[0]), ops.linear16_pack(ws[1], frag_in=(True, False)),
          ops.linear16_pack(ws[2], K1=K0, frag_in=(False, True)), ops.linear16_pack(ws[3], frag_in=(True, False))]
    f1, f2, f3 = (ops.FragBuffer(M, W, DEV) for _ in range(3))
    ops.linear16(x0, *pf[0], bs[0], relu=True, out=f1)
    assert torch.equal(f1.rows(), h1)                              # same kernel arithmetic, only the store differs
    ops.linear16(f1, *pf[1], bs[1], relu=True, out=f2)
    ops.linear16(x0, *pf[2], bs[2], relu=True, x2=f2, out=f3)
    yf = ops.linear16(f3, *pf[3], bs[3])
    _check(f2.rows(), h1, ws[1], bs[1], True)
    _check(f3.rows(), torch.cat([x0, f2.rows()], 1), ws[2], bs[2], True)
    _check(yf, f3.rows(), ws[3], bs[3], False)
    assert (yf - yr).abs().max().item() <= 2e-5 * yr.abs().max().item()
    with pytest.raises(ValueError):
        ops.linear16(f1, *pk[1], bs[1], relu=True, out=ops.FragBuffer(M, W + 32, DEV))
    with pytest.raises(ValueError):
        ops.FragBuffer(M, 100, DEV)


@pytest.mark.parametrize("L,M_rays,tau", [(7, 300, 20.0), (7, 37, 2000.0), (4, 129, 20.0)])
def test_linear16_recomputes_the_anerf_density_inputs_from_the_encoder_table(L, M_rays, tau):
    """danbo_linear16_fwd_enc: the first and the skip layer of the A-NeRF trunk on the encoder's compact table (the cutoff PE
    recomputed per k-step in the kernel) against the same layers on the 24 (1 + 2 L) + 72 rows k_anerf_encode writes, and
    against float64 on those rows; the compact table against the rows it stands for"""
    from core import hip_ops as ops
    from core.utils import synthetic as syn
    g = torch.Generator(device="cpu").manual_seed(L * 1000 + M_rays)
    S, G, W = 12, 1, 448
    scene = syn.make_scene(n_poses=1, H=32, W=32, n_views=1, pose_seed=3)
    ro, rd = scene["rays"][0]
    sel = np.linspace(0, ro.shape[0] - 1, M_rays).astype(np.int64)
    rays_o, rays_d = torch.tensor(ro[sel], device=DEV), torch.tensor(rd[sel], device=DEV)
    skts = torch.tensor(scene["skts"][:1], device=DEV, dtype=torch.float32)
    z = (torch.linspace(1.5, 4.5, S)[None, :] + torch.rand(M_rays, S, generator=g) * 0.1).to(DEV).contiguous()
    align = torch.eye(4, device=DEV).repeat(24, 1, 1).contiguous()
    cutoff = (torch.rand(24, generator=g) * 0.3 + 0.35).to(DEV)
    n = M_rays * S
    in_ch = 24 * (1 + 2 * L) + 72
    x0, w_rows = ops.anerf_encode(rays_o, rays_d, skts, align, cutoff, tau, L, 0, n, z=z)
    table, w_tab = ops.anerf_encode_compact(rays_o, rays_d, skts, align, cutoff, tau, 0, n, z=z)
    assert torch.equal(w_rows, w_tab)
    t, yz = table[:, :96].view(n, 24, 4), table[:, 96:].view(n, 24, 2)
    assert torch.equal(t[:, :, 0] * t[:, :, 2], x0[:, :24])                             # (cutoff - v) w
    assert torch.equal(torch.cat([t[:, :, 3:4], yz], 2).reshape(n, 72), x0[:, in_ch - 72:])
    assert (torch.sin(t[:, :, 1].double()) * t[:, :, 2].double() - x0[:, 24:48].double()).abs().max().item() < 2e-7
    w0 = (torch.randn(W, in_ch, generator=g) / in_ch ** 0.5).to(DEV)
    w5 = (torch.randn(W, in_ch + W, generator=g) / (in_ch + W) ** 0.5).to(DEV)
    b0, b5 = (torch.randn(W, generator=g) * 0.1).to(DEV), (torch.randn(W, generator=g) * 0.1).to(DEV)
    h4 = torch.relu(torch.randn(n, W, generator=g)).to(DEV)
    f4 = ops.FragBuffer.from_rows(h4)
    # the ordinary layers on the rows
    r1, r5 = ops.FragBuffer(n, W, DEV), ops.FragBuffer(n, W, DEV)
    ops.linear16(x0, *ops.linear16_pack(w0), b0, relu=True, out=r1)
    ops.linear16(x0, *ops.linear16_pack(w5, K1=in_ch, frag_in=(False, True)), b5, relu=True, x2=f4, out=r5)
    # the fused layers on the table
    e1, e5 = ops.FragBuffer(n, W, DEV), ops.FragBuffer(n, W, DEV)
    ops.linear16_enc(table, *ops.linear16_pack_enc(w0, L), L, b0, relu=True, out=e1)
    ops.linear16_enc(table, *ops.linear16_pack_enc(w5, L), L, b5, relu=True, x2=f4, out=e5)
    _check(e1.rows(), x0, w0, b0, True, tol=4e-6)
    _check(e5.rows(), torch.cat([x0, h4], 1), w5, b5, True, tol=4e-6)
    d1 = (e1.rows() - r1.rows()).abs().max().item() / r1.rows().abs().max().item()
    d5 = (e5.rows() - r5.rows()).abs().max().item() / r5.rows().abs().max().item()
    print("fused encoder layers vs the layers on the encoder's rows: relative to the largest output", d1, d5)
    assert d1 < 5e-6 and d5 < 5e-6
