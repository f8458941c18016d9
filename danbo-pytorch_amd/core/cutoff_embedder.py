"""Positional-encoding descriptors (reference: core/cutoff_embedder.py).

`Embedder` / `CutoffEmbedder` keep the reference's constructor arguments, `out_dim`, the
`cutoff_dist` parameter and `tau` buffer (so A-NeRF checkpoints load) and the tau / frequency
schedules; the encoding itself is evaluated inside the HIP kernels (PE of the blended voxel
feature in k_pe_mlp, PE of the view direction in k_view_consts, PE of the pose in k_pose_layer).
"""
import torch
import torch.nn as nn

from .utils.skeleton_utils import SMPLSkeleton


class Embedder(nn.Module):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], blocks of input_dims."""

    def __init__(self, freq_schedule=False, init_alpha=0.0, **kwargs):
        super().__init__()
        if freq_schedule:
            raise NotImplementedError("freq_schedule is not used by any shipped config")
        self.kwargs = kwargs
        self.freq_schedule, self.init_alpha = freq_schedule, init_alpha
        d = kwargs['input_dims']
        self.num_freqs = kwargs['num_freqs']
        self.freq_bands = 2. ** torch.linspace(0., kwargs['max_freq_log2'], steps=self.num_freqs) \
            if self.num_freqs > 0 else torch.zeros(0)
        self.out_dim = d * ((1 if kwargs['include_input'] else 0) + 2 * self.num_freqs)

    def forward(self, inputs, **kwargs):
        raise RuntimeError("positional encodings are evaluated inside libdanbo_hip kernels")

    def update_threshold(self, global_step, tau_step, tau_rate, alpha_step, alpha_target):
        pass

    def update_tau(self, *args, **kwargs):
        pass

    def get_tau(self):
        return 0.0


class CutoffEmbedder(Embedder):
    """A-NeRF's sigmoid-windowed PE: weight 1 - sigmoid(tau * (dist - cutoff))."""

    def __init__(self, cutoff_dist=500 * 0.00035, std=0.1, normalize=False, dist_inputs=False, cutoff_inputs=False,
                 opt_cutoff=False, cutoff_dim=24, freq_schedule=False, init_alpha=0., cut_to_cutoff=False,
                 shift_inputs=False, **kwargs):
        super().__init__(**kwargs)
        if normalize or opt_cutoff:
            raise NotImplementedError("normalize_cutoff / opt_cutoff are not used by any shipped config")
        self.dist_inputs, self.cutoff_inputs = dist_inputs, cutoff_inputs
        self.cut_to_cutoff, self.shift_inputs, self.cutoff_dim = cut_to_cutoff, shift_inputs, cutoff_dim
        self.cutoff_dist = nn.Parameter(torch.ones(cutoff_dim) * cutoff_dist, requires_grad=False)
        self.init_tau = 20.
        self.register_buffer('tau', torch.tensor(self.init_tau))

    def get_tau(self):
        return self.tau.item()

    def get_cutoff_dist(self):
        return self.cutoff_dist

    def update_threshold(self, global_step, tau_step, tau_rate, alpha_step, alpha_target):
        self.update_tau(global_step, tau_step, tau_rate)

    def update_tau(self, global_step, step, rate):
        # tau = 20 * rate^(step / (cutoff_step * 1000)), capped at 2000 (reference :221-223)
        self.tau = (self.init_tau * torch.ones_like(self.tau) * rate ** (global_step / float(step * 1000))).clamp(max=2000.)


def get_embedder(multires, i=0, input_dims=3, freq_schedule=False, init_alpha=0,
                 cutoff_kwargs={"cutoff": False}, skel_type=SMPLSkeleton, kc=False):
    if i == -1:
        return nn.Identity(), input_dims
    kw = dict(include_input=True, input_dims=input_dims, max_freq_log2=multires - 1, num_freqs=multires,
              log_sampling=True, skel_type=skel_type, freq_schedule=freq_schedule, init_alpha=init_alpha)
    if cutoff_kwargs["cutoff"]:
        kw.update({k: v for k, v in cutoff_kwargs.items() if k != "cutoff"})
        emb = CutoffEmbedder(**kw)
    else:
        emb = Embedder(**kw)
    return emb, emb.out_dim
