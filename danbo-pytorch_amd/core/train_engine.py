"""The training step on the HIP path: `danbo_train_step` (forward + losses + backward of one batch behind one C call,
csrc/k_train.hip) and `danbo_adam_step` on flat parameter / gradient buffers.

Reference: Trainer.train_batch (core/trainer.py:257-302) -- render, compute_loss (:348-394), loss.backward() and
optimizer.step() (:563-576) -- for the shipped DANBO structure.  What PyTorch does here: it owns the device memory
(the flat buffers every nn.Parameter / .grad / Adam moment is a VIEW of), draws the step's random numbers and provides the
stream; there is no autograd graph and no torch kernel on the hot path.

Flat layout: all trainable tensors back to back in the order of `_hip.TRAIN_TENSORS` (alpha_linear.bias moved right behind
feature_linear.bias: the two layers are evaluated as one 257-wide layer).  The gradient all-reduce of data-parallel training is
ONE collective on `flat_grad` with no packing copies.
"""
import ctypes
import math

import torch

from . import _hip

_P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())  # noqa: E731


def supported(args, caster):
    """-> None if the fused step covers this configuration, else the reason (the caller then uses the autograd path)"""
    net = caster.network
    if type(net).__name__ != 'DANBO':
        return f'network {type(net).__name__}'
    if args.loss_fn not in ('L1', 'MSE'):
        return f'loss_fn {args.loss_fn}'
    if getattr(args, 'reg_fn', None) not in (None, 'None') or getattr(args, 'weight_decay', None) is not None:
        return 'regulariser / weight decay'
    if getattr(args, 'finetune_light', False) or getattr(args, 'opt_pose', False):
        return 'finetune_light / opt_pose'
    # options the C step has no field for: the autograd path refuses them loudly (raycasters.RayCaster.render_rays_train),
    # so report them here instead of training with silently different semantics
    if getattr(args, 'lindisp', False):
        return 'lindisp'
    if float(getattr(args, 'ray_noise_std', 0.) or 0.) != 0.:
        return 'ray_noise_std'
    if getattr(args, 'density_type', 'relu') != 'relu':
        return f'density_type {args.density_type}'
    if not getattr(args, 'single_net', True):
        return 'single_net=False'
    if args.agg_type != 'sigmoid' or args.N_importance <= 0 or args.N_samples + args.N_importance > 256 or args.N_samples < 3:
        return 'sampling / aggregation settings'
    if net.voxel_pe_fn.num_freqs != 6 or net.W != 256 or net.D != 8 or list(net.skips) != [4]:
        return 'MLP shape'
    sd = dict(net.named_parameters())
    for n in _hip.TRAIN_TENSORS:
        if n not in sd and not (n.startswith('framecodes') and not net.use_framecode):
            return f'missing parameter {n}'
        if n in sd and not sd[n].requires_grad and n != 'graph_net.axis_scale':
            return f'{n} is frozen'
    extra = set(sd) - set(_hip.TRAIN_TENSORS)
    if extra:
        return f'parameters outside the fused step: {sorted(extra)}'
    return None


def adopt_adam_state(opt, params, offsets, flat_m, flat_v):
    """torch.optim.Adam's per-parameter state (what the checkpoint stores) becomes views of flat_m / flat_v; a state that was
    loaded from a checkpoint is copied in first.  -> the list of `step` tensors.

    `step`: ONE DISTINCT CPU tensor per parameter, as torch.optim.Adam keeps them.  (Round 3 let every entry reference one shared
    tensor; the sharing survives state_dict() / torch.save / load_state_dict, and a plain Adam that resumes from such a checkpoint
    -- the reference's loader, or this repo's autograd path -- then adds 1 per PARAMETER per step: bias corrections and the lr
    decay that reads state['step'] run ~43x too fast.)  The engine advances them together with one torch._foreach_add_."""
    count = None
    for p in params:
        st = opt.state.get(p)
        if st and 'step' in st:         # every parameter of the group has made the same number of steps
            count = float(st['step'])
            break
    steps = []
    for p, o in zip(params, offsets):
        k = p.numel()
        m, v = flat_m[o:o + k].view(p.shape), flat_v[o:o + k].view(p.shape)
        st = opt.state.get(p)
        if st and st['exp_avg'].data_ptr() != m.data_ptr():
            m.copy_(st['exp_avg'])
            v.copy_(st['exp_avg_sq'])
        step = torch.tensor(0. if count is None else count)
        opt.state[p] = dict(step=step, exp_avg=m, exp_avg_sq=v)
        steps.append(step)
    return steps


class DanboTrainEngine:
    def __init__(self, args, caster, optimizer):
        self.args, self.caster, self.opt = args, caster, optimizer
        net = self.net = caster.network
        dev = self.device = next(net.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError("the training step runs on the HIP path only: move the caster to a GPU first")
        params = dict(net.named_parameters())
        names = [n for n in _hip.TRAIN_TENSORS if n in params]
        order = list(names)
        order.remove('alpha_linear.bias')
        order.insert(order.index('feature_linear.bias') + 1, 'alpha_linear.bias')
        trainable = [n for n in order if params[n].requires_grad]
        frozen = [n for n in order if not params[n].requires_grad]
        # every tensor starts on a 16-byte boundary (vector loads in the kernels) except alpha_linear.bias, which must follow
        # feature_linear.bias immediately
        self.offsets, off = {}, 0
        for n in trainable + frozen:
            if n != 'alpha_linear.bias':
                off = (off + 3) // 4 * 4
            self.offsets[n] = off
            off += params[n].numel()
            if n == trainable[-1]:
                self.n_train = off
        total = off
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(self.n_train, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(self.n_train, device=dev, dtype=torch.float32)
        self.params = {n: params[n] for n in order}
        with torch.no_grad():
            for n, p in self.params.items():
                o, k = self.offsets[n], p.numel()
                self.flat_p[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + k].view(p.shape)
                p.grad = self.flat_g[o:o + k].view(p.shape)
        self.trainable = trainable
        self._adopt_optimizer_state()
        self.t = self._optimizer_step_count()
        self._buffers = {}
        self._ws = None
        self._rng_state, self._rng_seed = None, None     # danbo_random_draws' device-side state (see _rng)
        self._model_struct = None
        self.graph = None           # (key, CUDAGraph, static inputs, outputs)
        self.outputs_static = False
        self.generation = 0         # forward_backward calls so far: a replayed graph's outputs are STATIC buffers, valid until the next call
        self.use_graph = True
        self.fixed_draws = None     # dict(t_rand, u_rand, noise_c, noise_f) replaces the step's random draws (parity tests)

    # ------------------------------------------------------------------ optimizer state as views of the flat moments
    def _optimizer_step_count(self):
        for p in self.params.values():
            st = self.opt.state.get(p)
            if st and 'step' in st:
                return int(float(st['step']))
        return 0

    def _adopt_optimizer_state(self):
        self._param_list = list(self.params.values())
        self._step_tensors = adopt_adam_state(self.opt, [self.params[n] for n in self.trainable],
                                              [self.offsets[n] for n in self.trainable], self.flat_m, self.flat_v)

    # ------------------------------------------------------------------ model description for the C side
    def _model(self):
        if self._model_struct is not None:
            return self._model_struct
        net, args = self.net, self.args
        m = _hip.DanboTrainModel()
        for i, n in enumerate(_hip.TRAIN_TENSORS):
            if n in self.params:
                o = self.offsets[n]
                m.p[i] = self.flat_p.data_ptr() + 4 * o
                m.g[i] = self.flat_g.data_ptr() + 4 * o
        m.g_flat, m.n_flat = self.flat_g.data_ptr(), self.flat_g.numel()
        gl, pl = net.graph_net.layers, net.prob_linears.layers
        keep = self._buffers
        keep['adj0'] = gl[0].adj.detach().float().reshape(24, 24).contiguous()
        keep['adj1'] = gl[1].adj.detach().float().reshape(24, 24).contiguous()
        keep['adja'] = pl[0].adj.detach().float().reshape(24, 24).contiguous()
        from . import hip_ops
        hip_ops.check_smpl_adjacency(pl[0].adj)      # the forward of the step runs k_assign16 (SMPL neighbour table compiled in)
        keep['align'] = self.caster.transforms[0].to(self.device).float().contiguous()
        keep['init_scale'] = net.graph_net.init_scale.to(self.device).float().contiguous()
        m.g_adj0, m.g_adj1, m.a_adj = (keep[k].data_ptr() for k in ('adj0', 'adj1', 'adja'))
        m.align, m.init_scale = keep['align'].data_ptr(), keep['init_scale'].data_ptr()
        m.L_graph, m.graph_width = net.graph_pe_fn.num_freqs, gl[0].lin.weight.shape[-1]
        m.L_view, m.L_voxel = net.dirs_pe_fn.num_freqs, net.voxel_pe_fn.num_freqs
        m.ray_mode = 1 if net.pts_embedder.ray_tr_fn.encoder_name == "RLEncoder" else 0
        m.normalise = 1 if net.pts_embedder.view_input_fn.encoder_name == "VecNorm" else 0
        if net.use_framecode:
            m.n_codes, m.code_size = net.framecodes.codes.weight.shape
        m.view_ch = 3 * (1 + 2 * m.L_view) + (m.code_size if net.use_framecode else 0)
        m.use_volume_near_far = int(bool(getattr(self.caster, 'use_volume_near_far', False)))
        m.loss_mse, m.use_background = int(args.loss_fn == 'MSE'), int(bool(args.use_background))
        m.density_scale = float(args.density_scale)
        m.rgb_loss_coef, m.coarse_weight = float(args.rgb_loss_coef), float(args.coarse_weight)
        m.soft_softmax_coef = float(args.soft_softmax_loss_coef)
        m.vol_scale_penalty = float(args.vol_scale_penalty) if args.opt_vol_scale else 0.0
        self._model_struct = m
        return m

    # ------------------------------------------------------------------ one forward + backward
    def _launch(self, t, S, Sf, perturb, raw_noise_std, split=False):
        """t: dict of static input tensors; -> dict of outputs (device tensors).  split: only phase 1 (see _step_phase)"""
        m = self._model()
        R, G = t['rays_o'].shape[0], t['skts'].shape[0]
        dev = self.device
        B = m.density_scale
        # the step's random draws: ONE uniform and ONE normal generator launch (+ one scaling), carved into the four tensors the
        # C side wants contiguous: stratified offsets [R,S], inverse-CDF uniforms [R,Sf], density noise [R,S] and [R,S+Sf]
        rnd = {}
        if self.fixed_draws is not None:
            # parity hook: the caller supplies the step's random numbers (the reference's own draws, tests/golden/
            # danbo_perfcap_train_noise.npz): t_rand [R,S], u_rand [R,Sf] uniforms; noise_c [R,S], noise_f [R,S+Sf] ALREADY
            # multiplied by raw_noise_std * B as core/networks/nerf.py:316 forms them
            want = dict(t_rand=(R, S), u_rand=(R, Sf), noise_c=(R, S), noise_f=(R, S + Sf))
            for k, shp in want.items():
                v = self.fixed_draws.get(k)
                if v is not None:
                    if tuple(v.shape) != shp:
                        raise ValueError(f"fixed_draws[{k!r}]: shape {tuple(v.shape)}, expected {shp}")
                    if v.device != dev:
                        raise ValueError(f"fixed_draws[{k!r}] lives on {v.device}: move the draws to {dev} before the step (a copy "
                                         "inside a graph capture would be recorded into the graph)")
                    rnd[k] = v.float().contiguous()
            if perturb > 0. and ('t_rand' not in rnd or 'u_rand' not in rnd):
                raise ValueError("fixed_draws without t_rand / u_rand while perturb > 0: the step would run unperturbed")
            if raw_noise_std > 0. and ('noise_c' not in rnd or 'noise_f' not in rnd):
                raise ValueError("fixed_draws without noise_c / noise_f while raw_noise_std > 0: the step would run without noise")
        elif perturb > 0. or raw_noise_std > 0.:
            # ONE launch (danbo_random_draws: Philox4x32-10, its counter on the device, advanced by the kernel -- a replayed graph
            # draws fresh numbers by itself; torch's generators in a captured graph cost two fill launches per replay + a launch
            # per distribution, all in front of the step's first kernel): uniforms [R, S + Sf], normals * std * B [R, 2 S + Sf]
            # (ABI 7: launched BY the step, behind the fork of its prologue branches -- DanboTrainBatch.rng_*)
            n_u, n_n = (R * (S + Sf) if perturb > 0. else 0), (R * (2 * S + Sf) if raw_noise_std > 0. else 0)
            u = torch.empty(n_u, device=dev, dtype=torch.float32) if n_u else None
            nz = torch.empty(n_n, device=dev, dtype=torch.float32) if n_n else None
            rnd['_rng'] = (self._rng(), n_u, n_n, raw_noise_std * B)
            if u is not None:
                rnd['_u'] = u
                rnd['t_rand'], rnd['u_rand'] = u[:R * S].view(R, S), u[R * S:].view(R, Sf)
            if nz is not None:
                rnd['_n'] = nz
                rnd['noise_c'], rnd['noise_f'] = nz[:R * S].view(R, S), nz[R * S:].view(R, S + Sf)
        out = dict(rgb_map=(R, 3), disp_map=(R,), acc_map=(R,), alpha=(R, S + Sf), weights=(R, S + Sf), rgb0=(R, 3), disp0=(R,),
                   acc0=(R,), alpha0=(R, S), loss=(4,))
        out = {k: torch.empty(v, device=dev, dtype=torch.float32) for k, v in out.items()}
        out['counts'] = torch.empty(8, device=dev, dtype=torch.int32)
        chunk = R
        nbytes = self._c_workspace(m, R, G, S, Sf, chunk)
        if nbytes == 0:
            raise RuntimeError("the training step's workspace query rejected the model / batch shape")
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        bt = _hip.DanboTrainBatch(
            rays_o=_P(t['rays_o']), rays_d=_P(t['rays_d']), skts=_P(t['skts']), bones=_P(t['bones']), cyls=_P(t['cyls']),
            near_in=_P(t.get('near_in')), far_in=_P(t.get('far_in')), cam_idx=_P(t.get('cam_idx')), target=_P(t['target']),
            bgs=_P(t.get('bgs')), t_rand=_P(rnd.get('t_rand')), u_rand=_P(rnd.get('u_rand')), noise_c=_P(rnd.get('noise_c')),
            noise_f=_P(rnd.get('noise_f')), R=R, G=G, S=S, Sf=Sf, chunk=chunk)
        if '_rng' in rnd:
            state, n_u, n_n, std = rnd['_rng']
            bt.rng_state, bt.rng_uniform, bt.rng_normal = _P(state), _P(rnd.get('_u')), _P(rnd.get('_n'))
            bt.n_uniform, bt.n_normal, bt.normal_std = n_u, n_n, float(std)
        o = _hip.DanboTrainOut(**{k: _P(v) for k, v in out.items()})
        out['_keep'] = (rnd, bt, o)
        self._step_phase(out, 1 if split else 0)
        return out

    def _step_phase(self, out, phase):
        """phase 0: the whole step; 1: up to the pose-GNN adjoint (every gradient but the dense layers' final); 2: the rest"""
        _, bt, o = out['_keep']
        self._c_step(self._model(), bt, o, int(phase), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    # the two C entry points of the step (core/anerf_train_engine.py overrides them with danbo_anerf_train_*)
    def _c_workspace(self, m, R, G, S, Sf, chunk):
        return _hip.lib().danbo_train_workspace(ctypes.byref(m), R, G, S, Sf, chunk)

    def _c_step(self, m, bt, o, phase, stream):
        _hip.check(_hip.lib().danbo_train_step_phase(ctypes.byref(m), ctypes.byref(bt), ctypes.byref(o), _P(self._ws), self._ws.numel(),
                                                     phase, stream), "danbo_train_step")

    def _own_seed(self):
        """the seed of THIS rank's stream: torch's CUDA generator seed of the device, mixed with the RANK: ranks that were
        seeded alike -- a caller without a per-rank torch.manual_seed -- would otherwise draw the same stratified offsets and
        density noise for their different rays"""
        seed = int(torch.cuda.default_generators[self.device.index if self.device.index is not None else torch.cuda.current_device()].initial_seed())
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_rank() > 0:
            seed = (seed ^ (0x9E3779B97F4A7C15 * dist.get_rank())) & ((1 << 64) - 1)
        return seed

    def _rng(self):
        """the device-side generator state of danbo_random_draws (seed, counter, 0).  Seeded from torch's CUDA generator of this
        device the first time it is needed -- torch.manual_seed / torch.cuda.manual_seed before that (run_nerf.py: rank + 1) give
        every rank its own stream -- and again whenever that seed has CHANGED since; reseed() restarts it explicitly."""
        seed = self._own_seed()
        if self._rng_state is None:
            self._rng_state = torch.zeros(3, device=self.device, dtype=torch.int64)
            self._rng_seed = None
        if getattr(self, "_rng_restored_for", None) is not None:
            if int(torch.cuda.default_generators[self.device.index if self.device.index is not None else torch.cuda.current_device()].initial_seed()) == self._rng_restored_for:
                return self._rng_state               # a stream restored from a checkpoint: continue it
            self._rng_restored_for = None
        if seed != self._rng_seed:
            self.reseed(seed)
        return self._rng_state

    # ------------------------------------------------------------------ snapshot / restore of the whole training state
    def snapshot(self):
        """parameters, Adam moments and step count, the random stream: everything a step reads AND writes (bench.py restores it in
        front of every timed block, so that each block trains the same steps from the same weights -- ADVICE r4)"""
        return dict(p=self.flat_p.clone(), m=self.flat_m.clone(), v=self.flat_v.clone(), t=self.t,
                    steps=[float(x) for x in self._step_tensors],
                    rng=None if self._rng_state is None else self._rng_state.clone(), rng_seed=self._rng_seed)

    def restore(self, snap):
        self.flat_p.copy_(snap["p"])
        self.flat_m.copy_(snap["m"])
        self.flat_v.copy_(snap["v"])
        self.flat_g.zero_()
        self.t = snap["t"]
        for x, v in zip(self._step_tensors, snap["steps"]):
            x.fill_(v)
        if snap["rng"] is not None and self._rng_state is not None:
            self._rng_state.copy_(snap["rng"])
            self._rng_seed = snap["rng_seed"]
        torch._C._increment_version(self._param_list)      # packed weights of the eval path are keyed by the versions

    def rng_state_dict(self):
        """(seed, counter) of the step's random stream for a checkpoint (Trainer.save_nerf): a resumed run continues the stream
        instead of replaying the draws of step 0.  One host sync; None before the first draw."""
        if self._rng_state is None or self._rng_seed is None:
            return None
        st = self._rng_state.cpu()
        return dict(seed=int(self._rng_seed), counter=int(st[1]))

    def load_rng_state_dict(self, d):
        """continue the checkpointed stream.  The checkpoint holds RANK 0's (seed, counter); every rank draws the same number of
        values per step, so the counter is common to all of them, while the seed is each rank's own: rank 0 takes the saved one,
        rank r > 0 its rank-mixed seed of this run (with only rank 0 restored, the other ranks replayed the draws of
        step 0 after a resume)"""
        if not d:
            return
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.reseed(int(d["seed"]) if rank == 0 else self._own_seed())
        self._rng_state[1] = int(d["counter"])
        # the torch generator's seed of THIS process is whatever the resuming script set: keep the restored stream until it changes
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._rng_restored_for = int(torch.cuda.default_generators[dev].initial_seed())

    def reseed(self, seed):
        """restart the step's random stream at (seed, counter 0); the state tensor stays where it is (captured graphs hold it)"""
        seed = int(seed) & ((1 << 64) - 1)
        if self._rng_state is None:
            self._rng_state = torch.zeros(3, device=self.device, dtype=torch.int64)
        self._rng_seed = seed
        self._rng_state.copy_(torch.tensor([seed - (1 << 64) if seed >= (1 << 63) else seed, 0, 0], dtype=torch.int64))

    @staticmethod
    def _static_inputs(rays_o, rays_d, skts, bones, cyls, cam_idx, target, bgs, near_in, far_in, contiguous=True):
        # contiguous=False: rows (dim 0) may stay strided -- the graph path gathers them itself (danbo_gather_rows), e.g. the
        # trainer's per-pose slices x[::R / G] of the loader's per-ray tensors
        def f(x):
            if x is None:
                return None
            x = x.detach().float()
            return x.contiguous() if contiguous else x
        t = dict(rays_o=f(rays_o), rays_d=f(rays_d), skts=f(skts), bones=f(bones), cyls=f(cyls), target=f(target), bgs=f(bgs),
                 near_in=f(near_in), far_in=f(far_in))
        t['cam_idx'] = None if cam_idx is None else cam_idx.reshape(-1).to(torch.int64).contiguous()
        return {k: v for k, v in t.items() if v is not None}

    def forward_backward(self, rays_o, rays_d, skts, bones, cyls, cam_idx, target, bgs, S, Sf, perturb=0., raw_noise_std=0.,
                         near_in=None, far_in=None, split=False):
        """One batch: per-pose skts [G,24,4,4] / bones [G,24,3] / cyls [G,5]; per-ray everything else.  Gradients land in
        `flat_grad` (the parameters' .grad views); -> dict(rgb_map, ..., loss [4], counts [8])."""
        graphed = self.use_graph and self.fixed_draws is None    # supplied draws are per-step inputs: never captured into a graph
        self.generation += 1
        self.outputs_static = graphed        # a replayed graph writes the SAME output buffers every step; an eager step fresh ones
        if self.fixed_draws is None and (perturb > 0. or raw_noise_std > 0.):
            self._rng()                                          # (re)seeding copies host -> device: never inside a capture
        t = self._static_inputs(rays_o, rays_d, skts, bones, cyls, cam_idx if self.net.use_framecode else None, target, bgs,
                                near_in, far_in, contiguous=not graphed)
        if t.get('bgs') is not None and t['bgs'].numel() != t['target'].numel():
            t['bgs'] = t['bgs'].expand_as(t['target']).contiguous()
        if not graphed:
            out = self._launch(t, S, Sf, perturb, raw_noise_std, split)
            self._pending = (out, None) if split else None
            return out
        key = (tuple((k, tuple(v.shape)) for k, v in sorted(t.items())), S, Sf, float(perturb), float(raw_noise_std), bool(split))
        if self.graph is None or self.graph[0] != key:
            # the graph's static inputs are views of ONE flat buffer: a replay is preceded by a single gather of the caller's
            # tensors (danbo_gather_rows, one launch, rows of any stride; cam_idx travels as raw 32-bit words) instead of one
            # copy per tensor
            words = {k: v.numel() * (v.element_size() // 4) for k, v in t.items()}
            flat = torch.empty(sum((n + 63) // 64 * 64 for n in words.values()), device=self.device, dtype=torch.float32)
            static, spans, o = {}, [], 0
            for k, v in t.items():
                static[k] = flat[o:o + words[k]].view(v.dtype).view(v.shape)           # (contiguous whatever v's strides are)
                spans.append((k, o, words[k]))
                o += (words[k] + 63) // 64 * 64
            static['_flat'], static['_spans'] = flat, spans
            self._gather_inputs(static, t)
            cur = torch.cuda.current_stream()
            rng_before = None if self._rng_state is None else self._rng_state.clone()    # the warm-up's draws are given back
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):        # eager warm-up off the capture: lazy initialisations, workspace allocation
                w = self._launch(static, S, Sf, perturb, raw_noise_std, split)
                if split:
                    self._step_phase(w, 2)
            cur.wait_stream(side)
            # keep_graph: the captured hipGraph_t stays inspectable (raw_cuda_graph(); tests walk its edges: nothing may run beside K2)
            g = torch.cuda.CUDAGraph(keep_graph=True) if getattr(self, 'keep_graph', False) else torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = self._launch(static, S, Sf, perturb, raw_noise_std, split)
            g2 = None
            if split:                             # the weight-gradient GEMMs as their own graph: the all-reduce of the finished
                g2 = torch.cuda.CUDAGraph()       # gradients is launched between the two replays
                with torch.cuda.graph(g2):
                    self._step_phase(outs, 2)
            self.graph = (key, g, static, outs, g2)
            if rng_before is not None:            # a run's random stream does not depend on when (or how often) a graph was built
                self._rng_state.copy_(rng_before)
        _, g, static, outs, g2 = self.graph
        self._gather_inputs(static, t)
        g.replay()
        self._pending = (outs, g2) if split else None
        return outs

    @staticmethod
    def _gather_inputs(static, t):
        from . import hip_ops
        hip_ops.gather_rows(static['_flat'], [(t[k], off) for k, off, _ in static['_spans']])

    def finish_backward(self):
        """second half of a split step (forward_backward(..., split=True)): the dense layers' weight gradients"""
        outs, g2 = self._pending
        if g2 is not None:
            g2.replay()
        else:
            self._step_phase(outs, 2)
        self._pending = None

    def grad_buckets(self):
        """(finished after phase 1, finished after phase 2): the flat gradient of the pose GNN, the assignment net and the axis
        scales -- the first tensors of the flat layout, 7 of its 10 MB -- and the dense layers' (+ the frame codes, which sit
        behind them)"""
        cut = self.offsets['pts_linears.0.weight']
        return self.flat_g[:cut], self.flat_g[cut:self.n_train]

    # ------------------------------------------------------------------ optimizer
    def adam_step(self, lr, grad_scale=1.0):
        """torch.optim.Adam's update with the group's betas / eps on the flat buffers (core/raycasters.py:75)"""
        grp = self.opt.param_groups[0]
        b1, b2 = grp['betas']
        self.t += 1
        # the step's scalars are kernel ARGUMENTS (danbo_adam_step, ABI 2): the host may run any number of steps ahead of the GPU
        # (sync_stats=False) without a later step's bias corrections reaching an earlier step's launch
        _hip.check(_hip.lib().danbo_adam_step(_P(self.flat_p), _P(self.flat_g), _P(self.flat_m), _P(self.flat_v), self.n_train,
                                              float(lr), 1.0 - b1 ** self.t, math.sqrt(1.0 - b2 ** self.t), float(grad_scale),
                                              float(b1), float(b2), float(grp['eps']),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "danbo_adam_step")
        # torch.optim.Adam's per-parameter `step` entries (distinct CPU tensors, _adopt_optimizer_state): one fused in-place add
        torch._foreach_add_(self._step_tensors, 1.0)
        # the kernel wrote the parameters behind torch's back: bump their version counters, which the eval engine's packed weight
        # buffers (and their HIP graphs) are keyed on.  ONE call with the list: handed a single tensor, this torch iterates over
        # it (Tensor.__iter__ -> unbind) -- 43 parameters x unbind was 1.8 ms of host time per step, more than the step's GPU time.
        torch._C._increment_version(self._param_list)
