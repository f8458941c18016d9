"""Per-frame appearance codes (reference: core/networks/embedding.py:4-50).

Only parameter storage lives here; the lookup (or the mean code for idx < 0 in eval) happens
inside danbo_view_consts on the GPU (csrc/k_mlp.hip)."""
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

    def forward(self, idx, *args, **kwargs):
        raise RuntimeError("Optcodes lookups are fused into libdanbo_hip (danbo_view_consts); "
                           "call the owning NeRF/DANBO module instead")
