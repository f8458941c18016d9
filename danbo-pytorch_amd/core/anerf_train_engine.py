"""The A-NeRF (nerf_type = nerf) training step on the HIP path: `danbo_anerf_train_step` (forward + the two rgb losses + backward of
one batch behind one C call, csrc/k_anerf_train.hip) and `danbo_adam_step` on flat parameter / gradient buffers -- the A-NeRF
counterpart of core/train_engine.py, whose flat-buffer, Adam-state, random-stream and HIP-graph machinery it inherits.

Reference: Trainer.train_batch (core/trainer.py:257-302) with NeRF.forward (core/networks/nerf.py:107-122,176-209,222-279) and the
cutoff encoders (core/cutoff_embedder.py:151-214).  Every dense W-wide layer runs on k_linear16 both ways and k_dw16; the encoders,
the view branch, the colour head, the composites and every adjoint are this library's kernels: there is no torch / rocBLAS
kernel in the step (round 5 trained A-NeRF through autograd with library GEMMs for the heads and the view products).
"""
import ctypes

import torch

from . import _hip
from .train_engine import DanboTrainEngine, _P


def _names(net):
    D = len(net.pts_linears)
    names = ([f"pts_linears.{i}.weight" for i in range(D)] + [f"pts_linears.{i}.bias" for i in range(D)]
             + ["alpha_linear.weight", "alpha_linear.bias", "feature_linear.weight", "feature_linear.bias", "views_linears.0.weight",
                "views_linears.0.bias", "rgb_linear.weight", "rgb_linear.bias"])
    if net.use_framecode:
        names.append("framecodes.codes.weight")
    return names


def supported(args, caster):
    """-> None if danbo_anerf_train_step covers this configuration, else the reason (the caller then uses the autograd path)"""
    net = caster.network
    if type(net).__name__ != 'NeRF':
        return f'network {type(net).__name__}'
    if args.loss_fn not in ('L1', 'MSE'):
        return f'loss_fn {args.loss_fn}'
    if getattr(args, 'reg_fn', None) not in (None, 'None') or getattr(args, 'weight_decay', None) is not None:
        return 'regulariser / weight decay'
    if getattr(args, 'finetune_light', False) or getattr(args, 'opt_pose', False):
        return 'finetune_light / opt_pose'
    if getattr(args, 'lindisp', False):
        return 'lindisp'
    if float(getattr(args, 'ray_noise_std', 0.) or 0.) != 0.:
        return 'ray_noise_std'
    if getattr(args, 'density_type', 'relu') != 'relu':
        return f'density_type {args.density_type}'
    if not getattr(args, 'single_net', True):
        return 'single_net=False'
    if args.N_importance <= 0 or args.N_samples + args.N_importance > 256 or args.N_samples < 3:
        return 'sampling settings'
    try:
        cfg = net.engine_config()
    except Exception as e:       # an encoder combination the A-NeRF kernels do not cover
        return f'engine_config: {e}'
    D, W, VW = len(net.pts_linears), net.W, net.views_linears[0].weight.shape[0]
    if net.feature_linear.weight.shape[0] != W:
        return 'feature_linear width != W'          # (the reference: 2 view_W = W)
    skips = list(net.skips)
    if not (2 <= D <= _hip.ANERF_MAX_D) or W % 4 or W > 508 or VW % 4 or VW > 256 or len(net.views_linears) != 1:
        return 'MLP shape'
    if len(skips) > 1 or (skips and not 0 <= skips[0] <= D - 2):
        return f'skips {skips}'
    sd = dict(net.named_parameters())
    for n in _names(net):
        if n not in sd:
            return f'missing parameter {n}'
        if not sd[n].requires_grad:
            return f'{n} is frozen'
    extra = {n for n, p in sd.items() if p.requires_grad} - set(_names(net))
    if extra:
        return f'trainable parameters outside the fused step: {sorted(extra)}'
    if not torch.equal(net.pe_fn.cutoff_dist, net.dirs_pe_fn.cutoff_dist):
        return 'distance and view cutoffs differ'
    return None


class AnerfTrainEngine(DanboTrainEngine):
    def __init__(self, args, caster, optimizer):
        self.args, self.caster, self.opt = args, caster, optimizer
        net = self.net = caster.network
        dev = self.device = next(net.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError("the training step runs on the HIP path only: move the caster to a GPU first")
        params = dict(net.named_parameters())
        order = _names(net)
        # every tensor starts on a 16-byte boundary (vector loads / k_dw16's stores)
        self.offsets, off = {}, 0
        for n in order:
            off = (off + 3) // 4 * 4
            self.offsets[n] = off
            off += params[n].numel()
        self.n_train = total = off
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(total, device=dev, dtype=torch.float32)
        self.params = {n: params[n] for n in order}
        with torch.no_grad():
            for n, p in self.params.items():
                o, k = self.offsets[n], p.numel()
                self.flat_p[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + k].view(p.shape)
                p.grad = self.flat_g[o:o + k].view(p.shape)
        self.trainable = list(order)
        self._adopt_optimizer_state()
        self.t = self._optimizer_step_count()
        self._buffers = {}
        self._ws = None
        self._rng_state, self._rng_seed = None, None
        self._model_struct = None
        self.graph = None
        self.outputs_static = False
        self.generation = 0
        self.use_graph = True
        self.fixed_draws = None
        # tau as a DEVICE scalar at a fixed address: update_tau REPLACES the module's buffer every step (core/cutoff_embedder.py) and a
        # captured graph holds pointers -- forward_backward copies the current value in front of every step
        self._tau = torch.zeros(1, device=dev, dtype=torch.float32)

    def _model(self):
        if self._model_struct is not None:
            return self._model_struct
        net, args = self.net, self.args
        m = _hip.DanboAnerfTrainModel()
        cfg = net.engine_config()
        D = len(net.pts_linears)
        pp = lambda n: self.flat_p.data_ptr() + 4 * self.offsets[n]  # noqa: E731
        gp = lambda n: self.flat_g.data_ptr() + 4 * self.offsets[n]  # noqa: E731
        m.D, m.W, m.VW = D, net.W, net.views_linears[0].weight.shape[0]
        skips = list(net.skips)
        m.skip = skips[0] if skips else -1
        m.L, m.L_view = int(cfg["multires"]), int(cfg["multires_views"])
        for i in range(D):
            m.pts_w[i], m.pts_b[i] = pp(f"pts_linears.{i}.weight"), pp(f"pts_linears.{i}.bias")
            m.g_pts_w[i], m.g_pts_b[i] = gp(f"pts_linears.{i}.weight"), gp(f"pts_linears.{i}.bias")
        for f, n in (("alpha_w", "alpha_linear.weight"), ("alpha_b", "alpha_linear.bias"), ("feature_w", "feature_linear.weight"),
                     ("feature_b", "feature_linear.bias"), ("views_w", "views_linears.0.weight"), ("views_b", "views_linears.0.bias"),
                     ("rgb_w", "rgb_linear.weight"), ("rgb_b", "rgb_linear.bias")):
            setattr(m, f, pp(n))
            setattr(m, "g_" + f, gp(n))
        if net.use_framecode:
            m.n_codes, m.code_size = net.framecodes.codes.weight.shape
            m.codes, m.g_codes = pp("framecodes.codes.weight"), gp("framecodes.codes.weight")
        want = m.W + 72 * (1 + 2 * m.L_view) + m.code_size
        if net.views_linears[0].weight.shape[1] != want or net.pts_linears[0].weight.shape[1] != 24 * (1 + 2 * m.L) + 72:
            raise RuntimeError("A-NeRF layer widths do not match the cutoff encoders (reldist + reldir density inputs, relray view inputs)")
        m.g_flat, m.n_flat = self.flat_g.data_ptr(), self.flat_g.numel()
        keep = self._buffers
        keep['align'] = self.caster.transforms[0].to(self.device).float().contiguous()
        keep['cutoff'] = net.pe_fn.cutoff_dist.detach().to(self.device).float().contiguous()
        m.align, m.cutoff, m.tau = keep['align'].data_ptr(), keep['cutoff'].data_ptr(), self._tau.data_ptr()
        m.loss_mse, m.use_background = int(args.loss_fn == 'MSE'), int(bool(args.use_background))
        m.density_scale = float(args.density_scale)
        m.rgb_loss_coef, m.coarse_weight = float(args.rgb_loss_coef), float(args.coarse_weight)
        self._model_struct = m
        return m

    # ------------------------------------------------------------------ the C entry points
    def _c_workspace(self, m, R, G, S, Sf, chunk):
        return _hip.lib().danbo_anerf_train_workspace(ctypes.byref(m), R, G, S, Sf, chunk)

    def _c_step(self, m, bt, o, phase, stream):
        if phase == 2:          # a "split" step (data-parallel training): the whole step is phase 1, nothing is left for phase 2
            return
        _hip.check(_hip.lib().danbo_anerf_train_step(ctypes.byref(m), ctypes.byref(bt), ctypes.byref(o), _P(self._ws), self._ws.numel(), stream),
                   "danbo_anerf_train_step")

    def forward_backward(self, *a, **k):
        tau, tau_v = float(self.net.pe_fn.tau), float(self.net.dirs_pe_fn.tau)
        if tau != tau_v:
            raise NotImplementedError("pe_fn.tau != dirs_pe_fn.tau: the shared cutoff weight no longer applies")
        self._tau.copy_(self.net.pe_fn.tau.detach().reshape(1).to(self.device, torch.float32))
        out = super().forward_backward(*a, **k)
        c = out.get('counts')      # the A-NeRF step has no row counters (every sample is evaluated): zeros, once per static buffer
        if c is not None and (not self.outputs_static or getattr(self, '_counts_graph', None) is not self.graph):
            c.zero_()
            self._counts_graph = self.graph
        return out

    def grad_buckets(self):
        """(finished after phase 1, finished after phase 2): the whole flat gradient is final after the one phase of this step"""
        return self.flat_g[:self.n_train], self.flat_g[:0]

    def workspace_view(self, R, G, S, Sf):
        """device views of the last step's sampling decisions (z_coarse [R,S], z_fine [R,Sf], z_sorted, order [R,S+Sf])"""
        v = _hip.DanboTrainView()
        _hip.check(_hip.lib().danbo_anerf_train_workspace_view(ctypes.byref(self._model()), R, G, S, Sf, R, _P(self._ws), ctypes.byref(v)),
                   "danbo_anerf_train_workspace_view")
        return v
