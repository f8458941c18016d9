"""TEST INFRASTRUCTURE: the differentiable DANBO forward recorded LAYER BY LAYER with torch / library GEMMs -- the comparison the
HIP operators of core/train_path.py (torch.ops.danbo.pose_volumes / assign_blend / pe_mlp) are tested against
(tests/test_gpu_training.py).  Until round 4 these routes lived inside the product as the fall-back of a network of another shape;
the product raises for such a network now.  Reference: gnn_backbone.py:567-629,683-704, nerf.py:176-209."""
import torch
import torch.nn.functional as F

from core import hip_ops as ops
from core import train_path


def pose_volumes(model, bones_g):
    gn = model.graph_net
    n = train_path.positional_encoding(train_path.axis_angle_to_rot6d(bones_g), model.graph_pe_fn.num_freqs)
    mask = torch.ones(1, 24, 1, device=n.device)
    mask[:, 0] = 0.
    n = n * mask
    last = len(gn.layers) - 1
    for i, l in enumerate(gn.layers):
        if hasattr(l, "adj_w"):  # graph conv: per-bone linear, weighted adjacency, shared bias
            out = torch.matmul(l.get_adjw(), torch.einsum("bkl,klj->bkj", n, l.lin.weight)) + l.bias
        else:
            out = torch.einsum("bkl,klj->bkj", n, l.weight) + l.bias
        if i == 0:
            out = out + out  # skip_gcn=False quirk: first layer doubled (gnn_backbone.py:698-699)
        n = F.relu(out) if i < last else out
    return n


def assignment_logits(model, part_feat):
    l0, l1, l2 = model.prob_linears.layers
    y = torch.einsum("bkl,klj->bkj", part_feat, l0.lin.weight)
    y = F.relu(torch.matmul(l0.get_adjw(), y) + l0.bias)
    y = F.relu(torch.einsum("bkl,klj->bkj", y, l1.weight) + l1.bias)
    return (torch.einsum("bkl,klj->bkj", y, l2.weight) + l2.bias)[..., 0]


def mlp(model, dens_in, view_in):
    lin = lambda l, x: F.linear(x, l.weight, l.bias)  # noqa: E731
    h = dens_in
    for i, l in enumerate(model.pts_linears):
        h = F.relu(lin(l, h))
        if i in model.skips:
            h = torch.cat([dens_in, h], -1)
    alpha = lin(model.alpha_linear, h)
    hv = F.relu(lin(model.views_linears[0], torch.cat([lin(model.feature_linear, h), view_in], -1)))
    return torch.cat([lin(model.rgb_linear, hv), alpha], -1)


def forward_train(model, inputs):
    """core.train_path.forward_train with every operator replaced by its layer-by-layer torch form (the in-volume cull and the
    factorised gather stay: torch.ops.danbo.bone_gather has a test of its own)"""
    pts = inputs["pts"].contiguous().float()
    R, S = pts.shape[:2]
    G = int(inputs.get("N_uniques", 1))
    skts, bones = inputs["skts"], inputs["bones"]
    skts_g = (skts if skts.shape[0] == G else skts[:: max(skts.shape[0] // G, 1)]).contiguous().float()
    bones_g = (bones if bones.shape[0] == G else bones[:: max(bones.shape[0] // G, 1)]).contiguous().float()
    align = inputs["align_transforms"].reshape(-1, 24, 4, 4)[0].contiguous().float().to(pts.device)
    rays_d = inputs["rays_d"].reshape(R, 3).contiguous().float()
    axis_scale = model.graph_net.axis_scale
    geo = ops.Geometry(rays_d, rays_d, skts_g, align, axis_scale.detach(), pts=pts)
    bits, lst, cnt = ops.bone_cull(geo, compact=True)
    n = int(cnt.item())
    rows = torch.sort(lst[:n]).values.contiguous()
    shared = inputs.get("shared")
    shared = shared if shared is not None else {}
    if "vols" not in shared:
        shared["vols"] = pose_volumes(model, bones_g)
    vols = shared["vols"]
    shifts = torch.arange(24, device=pts.device, dtype=torch.int32)
    part_feat = torch.ops.danbo.bone_gather(vols, axis_scale, pts, skts_g, align, rows)
    logits = assignment_logits(model, part_feat)
    valid_rows = ((bits[rows.long()].unsqueeze(-1) >> shifts) & 1).float()
    p_rows = (torch.sigmoid(logits) * 1.002 - 0.001) * valid_rows
    h = (part_feat * p_rows[..., None]).sum(-2)
    if "vin" not in shared:
        shared["vin"] = train_path.view_inputs(model, rays_d, skts_g, inputs.get("cam_idxs"), R // G)
    vin = shared["vin"]
    ray_of_row = (rows // S).long()
    L = model.voxel_pe_fn.num_freqs
    pe = train_path.positional_encoding
    if "raw_empty" in shared:
        raw_empty = shared["raw_empty"]
        raw_rows = mlp(model, pe(h, L), vin[ray_of_row])
    else:
        pe_empty = pe(torch.zeros(1, h.shape[1], device=pts.device), L).expand(R, -1)
        raw_both = mlp(model, torch.cat([pe(h, L), pe_empty], 0), torch.cat([vin[ray_of_row], vin], 0))
        raw_rows, raw_empty = raw_both[:n], raw_both[n:]
        shared["raw_empty"] = raw_empty
    raw = raw_empty[:, None, :].expand(R, S, 4).reshape(R * S, 4).index_copy(0, rows.long(), raw_rows)
    confd = torch.zeros(R * S, 24, device=pts.device).index_copy(0, rows.long(), logits)
    p_valid = torch.zeros(R * S, 24, device=pts.device).index_copy(0, rows.long(), p_rows).reshape(R, S, 24)
    all_valid = ((bits.unsqueeze(-1) >> shifts) & 1).float()
    return raw.reshape(R, S, 4), dict(confd=confd.reshape(R, S, 24), part_invalid=(1.0 - all_valid).reshape(R, S, 24), p_valid=p_valid)
