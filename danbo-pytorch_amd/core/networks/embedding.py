"""Per-frame appearance codes (reference: core/networks/embedding.py:4-50).

Inside the render / training path the lookup (or the mean code for idx < 0 in eval) happens in danbo_view_consts on the GPU
(csrc/k_mlp.hip) and the module is parameter storage.  Called on its own -- what a reference-side caller of Optcodes.forward
does -- it runs the library's stand-alone lookup kernel (danbo_optcodes_fwd, csrc/k_encoders.hip): forward only."""
import torch
import torch.nn as nn


class Optcodes(nn.Module):
    def __init__(self, n_codes, code_ch, idx_map=None, transform_code=False, mean=None, std=None):
        super().__init__()
        if transform_code or idx_map is not None:
            raise NotImplementedError("idx_map / transform_code are not used by any shipped config")
        self.n_codes, self.code_ch = n_codes, code_ch
        self.codes = nn.Embedding(n_codes, code_ch)
        self.init_parameters(mean, std)

    def init_parameters(self, mean=None, std=None):
        if mean is None:
            nn.init.xavier_normal_(self.codes.weight)
        elif std > 0.:
            nn.init.normal_(self.codes.weight, mean=mean, std=std)
        else:
            nn.init.constant_(self.codes.weight, mean)

    def mean_code(self):
        return self.codes.weight.mean(0)

    def forward(self, idx, t=None, *args, **kwargs):
        """reference embedding.py:17-39: idx [N, 1] row indices (clamped to the table with the reference's warning), idx < 0 everywhere
        in eval mode: the mean code, idx [N, 3] = (row, row, weight): their torch.lerp.  No gradient flows through this call: training
        differentiates the codes inside the fused step / `torch.ops.danbo` path."""
        from .. import hip_ops as ops
        if torch.is_grad_enabled() and self.codes.weight.requires_grad and self.training:
            raise RuntimeError("Optcodes.forward is forward-only here; the codes' gradient comes from the fused training step "
                               "(core/train_engine.py) or the autograd path (core/train_path.py)")
        w = self.codes.weight.detach()
        if not self.training and float(idx.max()) < 0:
            return ops.optcodes(w, idx[..., :1], 1)
        if idx.shape[-1] != 1:
            return ops.optcodes(w, idx[..., :3], 2)
        if float(idx.max()) > self.n_codes:
            print('Warning! Out-of-range index detected in Optcodes input. Clamp it to self.n_codes-1')
        return ops.optcodes(w, idx, 0)
