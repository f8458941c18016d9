import sys, os
sys.path.insert(0, "danbo-pytorch_amd")
import torch
from core import hip_ops as ops
torch.manual_seed(0)
dev = "cuda:0"
R = 200000
for (S, Sf) in [(32, 16), (16, 8), (48, 16), (64, 32), (96, 48)]:
    for pat in range(8):
        near = torch.rand(R, 1, device=dev) * 3 + 1
        far = near + torch.rand(R, 1, device=dev) * 2 + 1e-3
        t = (torch.arange(S, device=dev).float() + torch.rand(R, S, device=dev)) / S
        z = near + (far - near) * t
        w = torch.rand(R, S, device=dev)
        if pat == 1: w = (w > 0.9).float()
        if pat == 2: w = torch.zeros_like(w); w[:, 0] = 1
        if pat == 3: w = torch.zeros_like(w)
        if pat == 4: w = torch.zeros_like(w); w[:, -1] = 1
        if pat == 5: w = w * 1e-30
        if pat == 6: w = (w > 0.97).float() * 1e10
        if pat == 7: z = near + (far - near) * torch.sort(torch.rand(R, S, device=dev), -1).values; z[:, 1] = z[:, 0]
        for mode in range(3):
            u = torch.rand(R, Sf, device=dev)
            if mode == 1: u = (u * 4).floor() / 4           # many ties, exact 0
            if mode == 2: u = 1 - torch.rand(R, Sf, device=dev) * 1e-7   # ~1 (fp32 rounds to 1.0 sometimes)
            if mode == 2: u = u.clamp(max=1 - 2**-24)
            zs, zf, idx = ops.importance_samples(z, w, Sf, u)
            torch.cuda.synchronize()
            ok_range = bool(((idx >= 0) & (idx < S + Sf)).all())
            srt = torch.sort(idx.long(), -1).values
            ok_perm = bool((srt == torch.arange(S + Sf, device=dev)).all()) if ok_range else False
            ok_sorted = bool((zs[:, 1:] >= zs[:, :-1]).all())
            ok_fin = bool(torch.isfinite(zs).all() and torch.isfinite(zf).all())
            both = torch.cat([z, zf], 1)
            ok_gather = ok_range and bool((torch.gather(both, 1, idx.long()) == zs).all())
            if not (ok_range and ok_perm and ok_sorted and ok_fin and ok_gather):
                print("FAIL", S, Sf, "pat", pat, "mode", mode, ok_range, ok_perm, ok_sorted, ok_fin, ok_gather,
                      "idx min/max", int(idx.min()), int(idx.max()), flush=True)
print("done")
