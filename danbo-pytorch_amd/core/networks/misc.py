"""Per-bone parallel linear layer and volume-extent initialisation
(reference: core/networks/misc.py:129-183, 675-724)."""
import math

import numpy as np
import torch
import torch.nn as nn

_FUSED = ("{} is a parameter container: its arithmetic is fused into libdanbo_hip kernels "
          "(include/danbo_hip.h); there is no eager fallback")


class ParallelLinear(nn.Module):
    """weight [n_parallel, in, out], bias [1, n_parallel, out] -- one independent linear map per
    bone (einsum 'bkl,klj->bkj' in the reference)."""

    def __init__(self, n_parallel, in_feat, out_feat, share=False, bias=True):
        super().__init__()
        if share:
            raise NotImplementedError("share=True is not used by any shipped config")
        self.n_parallel, self.in_feat, self.out_feat, self.share = n_parallel, in_feat, out_feat, share
        self.weight = nn.Parameter(torch.empty(n_parallel, in_feat, out_feat))
        if bias:
            self.bias = nn.Parameter(torch.zeros(1, n_parallel, out_feat))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # kaiming-uniform(a=sqrt 5) on each bone's [out, in] view == U(+-1/sqrt(in)); zero bias
        bound = 1.0 / math.sqrt(self.in_feat)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.zero_()

    def forward(self, x):
        raise RuntimeError(_FUSED.format("ParallelLinear"))

    def extra_repr(self):
        return f"n_parallel={self.n_parallel}, in_features={self.in_feat}, out_features={self.out_feat}, " \
               f"bias={self.bias is not None}"


def init_volume_scale(base_scale, skel_profile, skel_type):
    """Half extents [J,3] of every bone volume from rest-pose proportions."""
    J = len(skel_type.joint_names)
    x = torch.full((J,), float(base_scale))
    knee, shoulder = float(skel_profile['knee_width'][0]), float(skel_profile['shoulder_width'][0])
    collar = knee  # the reference reads knee_width for the collar term (misc.py:691); kept for parity
    x[torch.as_tensor(skel_profile['leg_idxs'])] = knee * 0.5
    x[torch.as_tensor(skel_profile['torso_idxs'])] = shoulder * 0.70
    x[torch.as_tensor(skel_profile['head_idxs'])] = shoulder * 0.60
    x[torch.as_tensor(skel_profile['arm_idxs'])] = collar * 0.60
    z = torch.tensor(np.asarray(skel_profile['bone_lens_to_child'][0]).astype(np.float32)) * 0.8
    z[z < 0] = z.max()                      # end effectors grow freely
    z[torch.as_tensor(skel_profile['head_idxs'])] = z.max() * 1.1
    return torch.stack([x, x.clone(), z], dim=-1)
