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


# ---- visualisations of the bone assignment (reference core/networks/misc.py:620-673), used by NeRF.raw2outputs(render_confd /
# render_entropy): plain element-wise torch on whatever device the logits live on -- not part of the hot path.
# One colour per SMPL joint: the CSS colours red, blue, yellow, magenta, green, indigo, darkorange, cyan, pink, yellowgreen,
# rosybrown, coral, chocolate, bisque, gold, yellowgreen, aquamarine, deepskyblue, navy, orchid, maroon, sienna, olive, lightgreen.
_JOINT_COLOURS_HEX = ("ff0000 0000ff ffff00 ff00ff 008000 4b0082 ff8c00 00ffff ffc0cb 9acd32 bc8f8f ff7f50 d2691e ffe4c4 ffd700 9acd32 "
                      "7fffd4 00bfff 000080 da70d6 800000 a0522d 808000 90ee90").split()


def joint_colours(device=None):
    rgb = [[int(h[i:i + 2], 16) / 255.0 for i in (0, 2, 4)] for h in _JOINT_COLOURS_HEX]
    return torch.tensor(rgb, dtype=torch.float32, device=device)


def get_confidence_rgb(confd, encoded=None):
    """colour of the bone with the largest assignment logit: confd [..., 24] -> [..., 3]"""
    return joint_colours(confd.device)[confd.argmax(dim=-1)]


def get_entropy_rgb(confd, encoded=None, eps=1e-7):
    """blue (one bone owns the sample) -> red (uniform over the bones) by the entropy of softmax(confd) relative to log(24)"""
    prob = torch.softmax(confd, dim=-1)
    ent = -(prob * (prob + eps).log()).sum(-1)
    ratio = (ent / math.log(confd.shape[-1]))[..., None]
    start = torch.tensor([0., 0., 1.], device=confd.device)
    end = torch.tensor([1., 0., 0.], device=confd.device)
    return torch.lerp(start.expand_as(ratio.expand(*ratio.shape[:-1], 3)), end.expand_as(ratio.expand(*ratio.shape[:-1], 3)), ratio)
