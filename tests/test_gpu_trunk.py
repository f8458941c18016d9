"""GPU tests of the fused trunk of the training step (csrc/k_mlp16.hip TRAIN instantiation, csrc/k_mlp16_bwd.hip) through the C
ABI: danbo_trunk_pack / danbo_trunk_fwd / danbo_trunk_bwd against an fp64 torch evaluation of the reference's network
(core/networks/nerf.py:176-209: pts_linears with the skip after layer 4, alpha / feature / views / rgb_linear) on random rows."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W, VW, IN = 256, 128, 195


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def frag_to_rows(buf, rows, C):
    """fragment order [rows/16][C/32][2][64 lanes][4] -> [rows, C]: lane = n + 16 q holds columns 32 s + 16 h + 4 q + i of row 16 g + n"""
    g = rows // 16
    x = buf[:g * 16 * C].view(g, C // 32, 2, 4, 16, 4)          # g, s, h, q, n, i
    return x.permute(0, 4, 1, 2, 3, 5).reshape(rows, C)


def lane_words(buf, rows):
    """per-lane words of a row group, [rows/16][lane = n + 16 q] -> numpy [rows, q]"""
    return buf[:rows * 4].view(rows // 16, 4, 16).permute(0, 2, 1).reshape(rows, 4).contiguous().cpu().numpy()


def pe_columns():
    from core import _hip
    return np.array([_hip.lib().danbo_trunk_pe_column(k) for k in range(224)])


class Net:
    """random weights in the reference's layouts, scaled like a trained network's (activations O(1))"""
    def __init__(self, seed, view_ch=155, scale=1.0):
        g = torch.Generator(device="cpu").manual_seed(seed)
        r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(DEV)   # noqa: E731
        self.view_ch = view_ch
        Ks = [IN, W, W, W, W, IN + W, W, W]
        self.pts_w = [r(W, K, k=scale * (2.0 / K) ** 0.5) for K in Ks]
        self.pts_b = [r(W, k=0.1) for _ in Ks]
        self.alpha_w, self.alpha_b = r(1, W, k=0.1), r(1, k=0.1)
        self.feature_w, self.feature_b = r(W, W, k=(1.0 / W) ** 0.5), r(W, k=0.1)
        self.views_w, self.views_b = r(VW, W + view_ch, k=(2.0 / (W + view_ch)) ** 0.5), r(VW, k=0.1)
        self.rgb_w, self.rgb_b = r(3, VW, k=0.1), r(3, k=0.1)

    def struct(self):
        from core import _hip
        self.packed = torch.empty(150 * 32768, dtype=torch.uint8, device=DEV)
        self.wfv = torch.empty(VW * W, device=DEV)
        self.b_eff = torch.empty(VW, device=DEV)
        self.wmax = torch.zeros(16, device=DEV)
        self.winv = torch.zeros(16, device=DEV)
        w = _hip.DanboTrunkWeights()
        for i in range(8):
            w.pts_w[i], w.pts_b[i] = self.pts_w[i].data_ptr(), self.pts_b[i].data_ptr()
        for n in ("alpha_w", "alpha_b", "feature_w", "feature_b", "views_w", "views_b", "rgb_w", "rgb_b", "packed", "wfv", "b_eff", "wmax",
                  "winv"):
            setattr(w, n, getattr(self, n).data_ptr())
        w.view_ch = self.view_ch
        return w

    def forward64(self, h, vin, keep_graph=False):
        """fp64 reference of the network on rows h [n,15] with per-row view inputs vin [n, view_ch]; keeps every activation
        (keep_graph: h is a leaf and every pre-activation retains its gradient)"""
        d = lambda t: t.double()   # noqa: E731
        L = 6
        h = d(h)
        if keep_graph:
            h.requires_grad_(True)
        pe = torch.cat([h] + [f(h * 2.0 ** l) for l in range(L) for f in (torch.sin, torch.cos)], 1)   # [n,195]
        ys, zs, x = [], [], pe
        for l in range(8):
            z = (torch.cat([pe, x], 1) if l == 5 else x) @ d(self.pts_w[l]).t() + d(self.pts_b[l])
            if keep_graph:
                z.retain_grad()
            x = torch.relu(z)
            ys.append(x)
            zs.append(z)
        alpha = x @ d(self.alpha_w).t() + d(self.alpha_b)
        feat = x @ d(self.feature_w).t() + d(self.feature_b)
        pre_v = torch.cat([feat, d(vin)], 1) @ d(self.views_w).t() + d(self.views_b)
        if keep_graph:
            pre_v.retain_grad()
        hv = torch.relu(pre_v)
        rgb = hv @ d(self.rgb_w).t() + d(self.rgb_b)
        return dict(pe=pe, ys=ys, zs=zs, hv=hv, pre_v=pre_v, h=h, raw=torch.cat([rgb, alpha], 1))


def make_rows(net, R, S, Sf, n_c, n_f, seed):
    """a step's row tables: R empty-space rows, n_c coarse and n_f importance in-volume rows with random samples / features"""
    from core import _hip
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = R + n_c + n_f
    cap = n + 37
    pad = (cap + 127) // 128 * 128 + 128
    t = {}
    t["cnt"] = torch.zeros(8, dtype=torch.int32, device=DEV)
    samp_c = torch.sort(torch.randperm(R * S, generator=g)[:n_c]).values
    samp_f = torch.sort(torch.randperm(R * Sf, generator=g)[:n_f]).values
    rs = torch.full((cap,), -7, dtype=torch.int32)
    rs[R:R + n_c] = samp_c.int()
    rs[R + n_c:n] = samp_f.int()
    t["row_sample"] = rs.to(DEV)
    h = torch.randn(cap, 16, generator=g) * 0.7
    h[:, 15] = 123.0                      # the pad slot (q in the training step): must not leak into the encoding
    t["h_rows"] = h.to(DEV)
    t["vin"] = torch.randn(R, net.view_ch, generator=g).to(DEV)
    for k, shape in dict(y=(8, pad * 256), pe=(pad * 224,), hv=(pad * 128,), raw_rows=(cap, 4), raw_c=(R * S, 4), raw_f=(R * Sf, 4),
                         raw_empty=(R, 4), dz=(8, pad * 256), dpre_v=(pad * 128,), d_alpha4=(cap, 4), d_h=(cap, 16), maxabs=(16,),
                         d_raw_rows=(cap, 4)).items():
        t[k] = torch.zeros(shape, device=DEV)
    t["relu"] = torch.zeros(8, pad * 4, dtype=torch.int64, device=DEV)
    t["hv_bits"] = torch.zeros(pad * 4, dtype=torch.int32, device=DEV)
    t["row_ray"] = torch.full((cap,), -1, dtype=torch.int32, device=DEV)
    t.update(R=R, S=S, Sf=Sf, n=n, n_c=n_c, n_f=n_f, cap=cap, pad=pad)
    return t


def rows_struct(t, cview):
    from core import _hip
    r = _hip.DanboTrunkRows()
    for k in ("cnt", "row_sample", "h_rows", "y", "pe", "relu", "hv", "hv_bits", "raw_rows", "raw_c", "raw_f", "raw_empty", "row_ray", "dz",
              "dpre_v", "d_alpha4", "d_h", "maxabs", "d_raw_rows"):
        setattr(r, k, t[k].data_ptr())
    r.cview = cview.data_ptr()
    if "d_raw_c" in t:
        r.d_raw_c, r.d_raw_f = t["d_raw_c"].data_ptr(), t["d_raw_f"].data_ptr()
    r.R, r.S, r.Sf, r.rows_cap, r.rows_pad = t["R"], t["S"], t["Sf"], t["cap"], t["pad"]
    return r


def run_forward(net, t):
    """pack + the two passes, with the counters advanced as the cull kernels of the step do"""
    from core import _hip
    l = _hip.lib()
    w = net.struct()
    _hip.check(l.danbo_trunk_pack(ctypes.byref(w), stream()), "trunk_pack")
    # per-ray view constants: cview = W_v[:, 256:] vin + b_eff
    cview = (t["vin"] @ net.views_w[:, W:].t() + net.b_eff).contiguous()
    r = rows_struct(t, cview)
    t["cnt"][0] = t["n_c"]
    _hip.check(l.danbo_trunk_fwd(ctypes.byref(w), ctypes.byref(r), 0, stream()), "trunk_fwd pass 0")
    t["cnt"][0] = t["n_c"] + t["n_f"]
    _hip.check(l.danbo_trunk_fwd(ctypes.byref(w), ctypes.byref(r), 1, stream()), "trunk_fwd pass 1")
    torch.cuda.synchronize()
    return w, r, cview


def reference_rows(net, t, keep_graph=False):
    R, S, Sf, n, n_c = t["R"], t["S"], t["Sf"], t["n"], t["n_c"]
    rs = t["row_sample"][:n].long()
    ray = torch.arange(n, device=DEV)
    ray[R:R + n_c] = rs[R:R + n_c] // S
    ray[R + n_c:] = rs[R + n_c:] // Sf
    h = t["h_rows"][:n, :15].clone()
    h[:R] = 0.0
    return ray, net.forward64(h, t["vin"][ray], keep_graph)


@pytest.mark.parametrize("R,S,Sf,n_c,n_f", [(96, 16, 8, 700, 333), (40, 32, 16, 0, 0), (130, 8, 4, 5, 250)])
def test_trunk_forward_matches_fp64_and_keeps_every_activation(R, S, Sf, n_c, n_f):
    net = Net(seed=R)
    t = make_rows(net, R, S, Sf, n_c, n_f, seed=n_c + 1)
    run_forward(net, t)
    n = t["n"]
    ray, ref = reference_rows(net, t)
    cnt = t["cnt"].cpu().tolist()
    first_f = R + n_c
    assert cnt[1:6] == [n_c, R + n_c, n_f, n, n_c + n_f] and cnt[6] == first_f // 128 * 128 and cnt[7] == n_f + first_f % 128
    assert torch.equal(t["row_ray"][:n].long(), ray)
    rows_p = (n + 15) // 16 * 16
    # raw: per row, and scattered to the dense tensors of the two passes / the rays' empty-space raw
    raw = t["raw_rows"][:n].double()
    scale = ref["raw"].abs().amax(0)
    err = ((raw - ref["raw"]).abs() / scale).max().item()
    print("raw: max |err| / channel max =", err)
    assert err < 2e-5
    rs = t["row_sample"][:n].long()
    assert torch.equal(t["raw_empty"], t["raw_rows"][:R])
    assert torch.equal(t["raw_c"][rs[R:first_f]], t["raw_rows"][R:first_f]) and torch.equal(t["raw_f"][rs[first_f:]], t["raw_rows"][first_f:n])
    # stored activations (fragment order), their sign bits, the encoding
    for l in range(8):
        y = frag_to_rows(t["y"][l], rows_p, 256)[:n].double()
        e = ((y - ref["ys"][l]).abs().max() / ref["ys"][l].abs().max()).item()
        assert e < 2e-5, (l, e)
        words = lane_words(t["relu"][l], rows_p).view(np.uint64)
        cols = np.arange(256)
        T_, q_, i_ = cols // 16, (cols % 16) // 4, cols % 4
        bits = ((words[:, q_] >> (4 * T_ + i_).astype(np.uint64)) & np.uint64(1)).astype(bool)[:n]
        assert np.array_equal(bits, (y > 0).cpu().numpy()), l
    hv = frag_to_rows(t["hv"], rows_p, 128)[:n].double()
    assert ((hv - ref["hv"]).abs().max() / ref["hv"].abs().max()).item() < 2e-5
    hb = lane_words(t["hv_bits"], rows_p).view(np.uint32)
    cols = np.arange(128)
    bits = ((hb[:, (cols % 16) // 4] >> (4 * (cols // 16) + cols % 4).astype(np.uint32)) & 1).astype(bool)[:n]
    assert np.array_equal(bits, (hv > 0).cpu().numpy())
    pe = frag_to_rows(t["pe"], rows_p, 224)[:n].double()
    colmap = pe_columns()
    live = colmap >= 0
    assert sorted(colmap[live].tolist()) == list(range(195))
    assert (pe[:, live] - ref["pe"][:, colmap[live]]).abs().max().item() < 1e-6
    # (the padding slots hold the encoding of a zero channel -- cos 0 = 1 --: the packed weights are zero there and the
    #  weight-gradient kernel drops those columns)


@pytest.mark.parametrize("R,S,Sf,n_c,n_f,gscale", [(96, 16, 8, 700, 333, 1e-6), (130, 8, 4, 5, 250, 3e-9), (64, 8, 4, 0, 0, 1e-4)])
def test_trunk_backward_matches_fp64_autograd(R, S, Sf, n_c, n_f, gscale):
    """d raw (dense gradients of the two passes + the rays' empty-space sums) -> dz_0 .. dz_7, d pre_v, d alpha, d h against
    torch.autograd in fp64 through the same network, at gradient magnitudes far below fp16's range"""
    from core import _hip
    net = Net(seed=R + 1)
    t = make_rows(net, R, S, Sf, n_c, n_f, seed=n_c + 3)
    g = torch.Generator(device="cpu").manual_seed(5)
    n, first_f = t["n"], R + n_c
    # per-row weights differ by orders of magnitude (an empty-space row sums a whole ray): the per-wavefront pre-scale has to cope
    row_gain = torch.exp(torch.randn(t["cap"], 1, generator=g) * 2.0)
    t["d_raw_c"] = (torch.randn(R * S, 4, generator=g) * gscale).to(DEV)
    t["d_raw_f"] = (torch.randn(R * Sf, 4, generator=g) * gscale).to(DEV)
    t["d_raw_rows"] = (torch.randn(t["cap"], 4, generator=g) * gscale * row_gain).to(DEV)
    w, r, cview = run_forward(net, t)
    r = rows_struct(t, cview)
    rs = t["row_sample"][:n].long()
    G = t["d_raw_rows"][:n].clone()
    G[R:first_f] = t["d_raw_c"][rs[R:first_f]]
    G[first_f:] = t["d_raw_f"][rs[first_f:]]
    _hip.check(_hip.lib().danbo_trunk_bwd(ctypes.byref(w), ctypes.byref(r), stream()), "trunk_bwd")
    torch.cuda.synchronize()
    ray, ref = reference_rows(net, t, keep_graph=True)
    (ref["raw"] * G.double()).sum().backward()
    rows_p = (n + 15) // 16 * 16
    assert torch.equal(t["d_raw_rows"][:n], G) and torch.equal(t["d_alpha4"][:n, 0], G[:, 3]) and float(t["d_alpha4"][:n, 1:].abs().max()) == 0.0
    worst = 0.0

    def group_err(a, b):
        """largest error relative to the largest reference entry of the same 16-row group (the unit that shares a pre-scale; the
        groups themselves differ by orders of magnitude)"""
        k = a.shape[0]
        pad = (-k) % 16
        ea = torch.nn.functional.pad((a - b).abs().amax(1), (0, pad)).view(-1, 16).amax(1)
        eb = torch.nn.functional.pad(b.abs().amax(1), (0, pad)).view(-1, 16).amax(1)
        return (ea / (eb + 1e-300)).max().item()
    for l in range(8):
        dz = frag_to_rows(t["dz"][l], rows_p, 256)[:n].double()
        refz = ref["zs"][l].grad
        e = group_err(dz, refz)
        print("layer", l, "group-relative", e, "tensor-relative", ((dz - refz).abs().max() / refz.abs().max()).item())
        worst = max(worst, e)
        assert e < 5e-5, (l, e)
        mx = float(t["maxabs"][l])
        assert mx >= float(dz.abs().max()) * (1 - 1e-6) and mx <= 64 * float(dz.abs().max()) + 1e-30, (l, mx, float(dz.abs().max()))
    dv = frag_to_rows(t["dpre_v"], rows_p, 128)[:n].double()
    e = group_err(dv, ref["pre_v"].grad)
    assert e < 1e-5, e
    dh = t["d_h"][R:n, :15].double()
    refh = ref["h"].grad[R:]
    if n > R:
        k0 = (-R) % 16          # align the groups with the kernel's (rows R .. n start inside a group)
        e = max(group_err(dh[:k0], refh[:k0]) if k0 else 0.0, group_err(dh[k0:], refh[k0:]) if n - R > k0 else 0.0)
        print("d h group-relative", e)
        worst = max(worst, e)
        assert e < 5e-5, e
        assert float(t["d_h"][R:n, 15].abs().max()) == 0.0
    print("worst per-row relative deviation", worst)


def test_pe_mlp_custom_op_forward_backward_and_opcheck():
    """torch.ops.danbo.pe_mlp (core/custom_ops.py): raw and the gradient of EVERY input that carries one -- h, the per-ray view
    inputs, the 24 parameter tensors -- against fp64 autograd through the reference's network; schema / fake-tensor opcheck"""
    from core import custom_ops  # noqa: F401
    net = Net(seed=7, view_ch=155)
    g = torch.Generator(device="cpu").manual_seed(11)
    n, R = 777, 50
    h = (torch.randn(n, 15, generator=g) * 0.7).to(DEV).requires_grad_(True)
    row_ray = torch.sort(torch.randint(0, R, (n,), generator=g)).values.int().to(DEV)
    vin = torch.randn(R, 155, generator=g).to(DEV).requires_grad_(True)
    params = [p.clone().requires_grad_(True) for p in net.pts_w + net.pts_b + [net.alpha_w, net.alpha_b, net.feature_w, net.feature_b,
                                                                                net.views_w, net.views_b, net.rgb_w, net.rgb_b]]
    raw = torch.ops.danbo.pe_mlp(h, row_ray, vin, params)
    G = (torch.randn(n, 4, generator=g) * 1e-5).to(DEV)
    (raw * G).sum().backward()
    # fp64 reference with the same leaves
    h64, vin64 = h.detach().double().requires_grad_(True), vin.detach().double().requires_grad_(True)
    net64 = Net(seed=7, view_ch=155)
    names = ["pts_w", "pts_b"]
    p64 = [p.detach().double().requires_grad_(True) for p in params]
    net64.pts_w, net64.pts_b = p64[:8], p64[8:16]
    net64.alpha_w, net64.alpha_b, net64.feature_w, net64.feature_b, net64.views_w, net64.views_b, net64.rgb_w, net64.rgb_b = p64[16:]
    pe = torch.cat([h64] + [f(h64 * 2.0 ** l) for l in range(6) for f in (torch.sin, torch.cos)], 1)
    x = pe
    for l in range(8):
        x = torch.relu((torch.cat([pe, x], 1) if l == 5 else x) @ p64[l].t() + p64[8 + l])
    alpha = x @ p64[16].t() + p64[17]
    feat = x @ p64[18].t() + p64[19]
    hv = torch.relu(torch.cat([feat, vin64[row_ray.long()]], 1) @ p64[20].t() + p64[21])
    ref = torch.cat([hv @ p64[22].t() + p64[23], alpha], 1)
    (ref * G.double()).sum().backward()
    assert ((raw.detach().double() - ref.detach()).abs() / ref.detach().abs().amax(0)).max().item() < 2e-5
    worst = 0.0
    for name, a, b in [("h", h.grad, h64.grad), ("vin", vin.grad, vin64.grad)] + [(f"param{i}", p.grad, q.grad) for i, (p, q) in enumerate(zip(params, p64))]:
        e = ((a.double() - b).abs().max() / (b.abs().max() + 1e-300)).item()
        worst = max(worst, e)
        assert e < 1e-4, (name, e)
    print("pe_mlp op: worst gradient deviation relative to the tensor's max", worst)
    torch.library.opcheck(torch.ops.danbo.pe_mlp, (h.detach(), row_ray, vin.detach(), [p.detach() for p in params]),
                          test_utils=("test_schema", "test_faketensor"))
