"""CPU-only tests of everything that is not a GPU kernel launch:
  * the C-ABI library loads and exports every symbol include/danbo_hip.h declares,
  * the kernel bodies that are plain scalar C++ (csrc/sample_math.hpp), host-compiled, agree
    with the numpy oracle / golden vectors (mask bit-exact),
  * the core.networks / core.raycasters module surface: names, state_dict keys and shapes,
    checkpoint round trip, config parsing, and the hard failure without a GPU.
"""
import copy
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import danbo_oracle as o
from helpers import ROOT, golden, oracle_for, max_err

F = ctypes.POINTER(ctypes.c_float)


def fp(a):
    return a.ctypes.data_as(F)


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from core import _hip
    hdr = open(os.path.join(ROOT, "include", "danbo_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|long)\s+(danbo_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 16
    lib = _hip.lib()                                  # loads without a GPU
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/danbo_hip.h but not exported"
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    assert lib.danbo_abi_version() == 9
    # argument counts of the ctypes table match the header
    for name in declared:
        m = re.search(r"(?:int|size_t|long)\s+" + name + r"\s*\((.*?)\);", hdr, flags=re.S)
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S).strip()
        n_args = 0 if body in ("void", "") else body.count(",") + 1
        assert n_args == len(_hip.SIGNATURES[name]), (name, n_args, len(_hip.SIGNATURES[name]))


def test_invalid_arguments_are_rejected_without_touching_the_gpu():
    from core import _hip
    lib = _hip.lib()
    # R % G != 0 -> DANBO_EINVAL before any launch
    rc = lib.danbo_near_far_boxes(None, None, None, None, None, 10, 3, None, None, None)
    assert rc == -22
    assert lib.danbo_composite_fwd(None, None, None, 0, 8, 1.0, None, None, None, None, None, None, None) == -22


def test_ops_refuse_cpu_tensors():
    from core import hip_ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hip_ops.composite(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 3))


# ------------------------------------------------------------------ host-compiled kernel bodies
@pytest.fixture(scope="module")
def emu():
    d = os.path.join(ROOT, "tests", "host_emu")
    subprocess.check_call(["make", "-C", d], stdout=subprocess.DEVNULL)
    return ctypes.CDLL(os.path.join(d, "libdanbo_emu.so"))


def test_emu_transform_cull_gather_vs_golden(emu):
    g = golden("danbo_stages")
    rb = np.ascontiguousarray(g["ray_batch"])
    R, S = g["z_coarse"].shape
    ro, rd = np.ascontiguousarray(rb[:, 0:3]), np.ascontiguousarray(rb[:, 3:6])
    z = np.zeros((R, S), np.float32)
    emu.emu_coarse_z(fp(np.ascontiguousarray(g["near"][:, 0])), fp(np.ascontiguousarray(g["far"][:, 0])), R, S, fp(z))
    assert np.array_equal(z, g["z_coarse"])
    orc, cfg, sd, rest = oracle_for(g)
    bits = np.zeros(R * S, np.uint32)
    pts_t = np.zeros((R, S, 24, 3), np.float32)
    pf = np.zeros((R, S, 24, 15), np.float32)
    emu.emu_cull_gather(fp(ro), fp(rd), fp(z), R, S, 2, fp(np.ascontiguousarray(g["skts"])), fp(orc.align),
                        fp(np.ascontiguousarray(sd["graph_net.axis_scale"])), fp(np.ascontiguousarray(g["volumes"])),
                        bits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), fp(pts_t), fp(pf))
    assert np.array_equal(pts_t, g["pts_t"])                       # bit-exact vs the reference
    valid = ((bits.reshape(R, S)[..., None] >> np.arange(24, dtype=np.uint32)) & 1).astype(bool)
    assert np.array_equal(~valid, g["invalid"].astype(bool))
    assert max_err(pf, g["part_feat"]) < 5e-6


def test_emu_cylinder_and_boxes_vs_oracle(emu):
    from core.utils import synthetic as syn
    g = golden("danbo_surreal")
    orc, cfg, sd, rest = oracle_for(g)
    scene = syn.make_scene(n_poses=1, H=64, W=64, n_views=3, pose_seed=int(g["pose_seed"]),
                           rest_scale=cfg["rest_scale"], cam_dist=float(g["cam_dist"]))
    ro, rd = scene["rays"][int(g["view"])]
    R = len(ro)
    nr, fr, hit = np.zeros(R, np.float32), np.zeros(R, np.float32), np.zeros(R, np.int32)
    emu.emu_cylinder(fp(ro), fp(rd), fp(scene["cyls"]), R, 1, ctypes.c_float(0.), ctypes.c_float(1.), fp(nr), fp(fr),
                     hit.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    miss = hit == 0
    assert 50 < miss.sum() < 2000
    nr[miss] = np.float32(nr[~miss].astype(np.float64).mean())      # chunk-wide nan-mean
    fr[miss] = np.float32(fr[~miss].astype(np.float64).mean())
    # bit for bit the reference's tensors (round 5: torch.norm is an fma chain, sample_math.hpp norm2_torch / norm3_torch; the
    # chunk's nan-mean here is numpy's, as in the reference)
    assert np.array_equal(nr[~miss], g["cyl_near"][~miss, 0]) and np.array_equal(fr[~miss], g["cyl_far"][~miss, 0])
    assert max_err(nr, g["cyl_near"][:, 0]) < 3e-7 and max_err(fr, g["cyl_far"][:, 0]) < 3e-7
    emu.emu_boxes(fp(ro), fp(rd), fp(scene["skts"]), fp(orc.align), fp(np.ascontiguousarray(sd["graph_net.axis_scale"])),
                  R, 1, fp(nr), fp(fr))
    boxed = g["near"][:, 0] != g["cyl_near"][:, 0]
    assert boxed.sum() > 1000
    assert np.array_equal(nr[boxed], g["near"][boxed, 0]) and np.array_equal(fr[boxed], g["far"][boxed, 0])     # the box bounds: exact
    assert max_err(nr, g["near"][:, 0]) < 3e-7 and max_err(fr, g["far"][:, 0]) < 3e-7


def test_emu_importance_and_composite_vs_golden(emu):
    g = golden("danbo_stages")
    R, S = g["z_coarse"].shape
    Sf = int(g["N_importance"])
    zf, zs = np.zeros((R, Sf), np.float32), np.zeros((R, S + Sf), np.float32)
    idx = np.zeros((R, S + Sf), np.int32)
    emu.emu_importance(fp(np.ascontiguousarray(g["z_coarse"])), fp(np.ascontiguousarray(g["weights_coarse"])), R, S, Sf,
                       fp(zf), fp(zs), idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    assert max_err(zf, g["z_fine"]) < 5e-6 and max_err(zs, g["z_sorted"]) < 5e-6
    assert np.array_equal(idx.astype(np.int64), g["sorted_idxs"])
    rgb, disp, acc = np.zeros((R, 3), np.float32), np.zeros(R, np.float32), np.zeros(R, np.float32)
    w, al = np.zeros((R, S), np.float32), np.zeros((R, S), np.float32)
    rd = np.ascontiguousarray(g["ray_batch"][:, 3:6])
    emu.emu_composite(fp(np.ascontiguousarray(g["raw_coarse"])), fp(np.ascontiguousarray(g["z_coarse"])), fp(rd), R, S,
                      ctypes.c_float(1.0), None, fp(rgb), fp(disp), fp(acc), fp(w), fp(al))
    assert max_err(w, g["weights_coarse"]) < 2e-6 and max_err(rgb, g["rgb_coarse"]) < 2e-6
    assert max_err(acc, g["final_acc0"]) < 2e-6


# ------------------------------------------------------------------ module surface
def _build(cfg_file, extra=()):
    from core.config import parse_args
    from core.raycasters import create_raycaster
    from core.utils import synthetic as syn
    from core.utils.skeleton_utils import SMPLSkeleton
    args = parse_args(["--no_reload", *extra], config=os.path.join(ROOT, "danbo-pytorch_amd", "configs", cfg_file))
    scale = 0.714 if "surreal" in cfg_file else 0.48
    da = dict(skel_type=SMPLSkeleton, near=0., far=100., n_views=20, rest_pose=syn.rest_pose(scale), hwf=(64, 64, 80.))
    return args, create_raycaster(args, da)


@pytest.mark.parametrize("cfg_file,cfg_name", [("h36m_zju/danbo_base.txt", "danbo_base"),
                                               ("perfcap/danbo_fast.txt", "danbo_perfcap"),
                                               ("surreal/danbo_fast.txt", "danbo_surreal")])
def test_state_dict_names_and_shapes_match_reference(cfg_file, cfg_name):
    from core.utils import synthetic as syn
    args, (tr, te, start, grad_vars, opt, _) = _build(cfg_file)
    caster = te["ray_caster"]
    cfg = syn.model_config(cfg_name)
    ref = syn.make_state_dict(cfg, 0, 20, syn.rest_pose(cfg["rest_scale"]))   # loads strictly into the reference
    sd = caster.network.state_dict()
    assert set(sd) == set(ref)
    for k in ref:
        assert tuple(sd[k].shape) == ref[k].shape, k
    assert max_err(sd["graph_net.axis_scale"].numpy(), ref["graph_net.axis_scale"]) == 0.0
    assert np.array_equal(sd["graph_net.layers.0.adj"].numpy(), ref["graph_net.layers.0.adj"])
    n_par = sum(p.numel() for p in grad_vars)
    assert n_par == sum(v.size for k, v in ref.items() if not k.endswith(".adj"))
    assert caster.network_fine is caster.network                # single_net
    assert type(caster).__name__ == "GraphCaster"
    assert caster.use_volume_near_far == ("fast" in cfg_file)
    assert np.array_equal(caster.transforms[0].numpy(), o.bone_align_transforms(syn.rest_pose(cfg["rest_scale"])))


def test_reference_checkpoint_layout_round_trip(tmp_path):
    from core.utils import synthetic as syn
    args, (tr, te, *_rest) = _build("h36m_zju/danbo_base.txt")
    caster = te["ray_caster"]
    ref = syn.make_state_dict(syn.model_config("danbo_base"), 5, 20, syn.rest_pose(0.48))
    caster.network.load_state_dict({k: torch.tensor(v) for k, v in ref.items()}, strict=True)
    ck = caster.state_dict()
    # reference layout (raycasters.py:601-615): one dict per sub-network, single_net saved twice
    assert set(ck) == {"network_fn_state_dict", "network_fine_state_dict"}
    path = tmp_path / "000010.tar"
    torch.save({"global_step": 10, **ck}, path)
    args2, (tr2, te2, *_r) = _build("h36m_zju/danbo_base.txt")
    c2 = te2["ray_caster"]
    c2.load_state_dict(torch.load(path))
    for k, v in c2.network.state_dict().items():
        assert torch.equal(v, torch.tensor(ref[k])), k


def test_config_files_parse_to_the_shipped_values():
    from core.config import parse_args
    d = os.path.join(ROOT, "danbo-pytorch_amd", "configs")
    a = parse_args([], config=os.path.join(d, "h36m_zju/danbo_base.txt"))
    assert (a.nerf_type, a.N_samples, a.N_importance, a.multires_voxel, a.gcn_fc_D, a.chunk) == ("danbo", 96, 48, 6, 1, 4096)
    assert a.use_volume_near_far is False and a.opt_framecode is True and a.loss_fn == "L1"
    b = parse_args(["--N_samples", "48"], config=os.path.join(d, "perfcap/danbo_fast.txt"))
    assert (b.nerf_type, b.view_type, b.ray_tr_type, b.N_samples, b.vol_scale_penalty) == ("graph", "relray", "root_local", 48, 1e-3)   # the reference file's value (its README trains PerfCap with --vol_scale_penalty 0.0001)
    c = parse_args([], config=os.path.join(d, "h36m_zju/anerf_base.txt"))
    assert (c.nerf_type, c.netwidth, c.use_cutoff, c.multires) == ("nerf", 448, True, 7)


def test_cutoff_embedder_tau_schedule():
    from core.cutoff_embedder import get_embedder
    emb, dim = get_embedder(7, 0, input_dims=24, cutoff_kwargs=dict(cutoff=True, cutoff_dist=0.5, cutoff_dim=24,
                                                                    cutoff_inputs=True, cut_to_cutoff=True,
                                                                    shift_inputs=True))
    assert dim == 360 and emb.get_tau() == 20.0
    emb.update_tau(250000, 250, 10.0)
    assert abs(emb.get_tau() - 200.0) < 1e-3
    emb.update_tau(10 ** 7, 250, 10.0)
    assert emb.get_tau() == 2000.0
    assert set(emb.state_dict()) == {"cutoff_dist", "tau"}


def test_forward_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    args, (tr, te, *_r) = _build("h36m_zju/danbo_base.txt")
    caster = te["ray_caster"].eval()
    R = 8
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        caster(torch.zeros(R, 11), N_samples=8, kp_batch=torch.zeros(R, 24, 3), skts=torch.eye(4).expand(R, 24, 4, 4),
               cyls=torch.ones(R, 5), bones=torch.zeros(R, 24, 3), cams=None, N_importance=4)


def test_unsupported_variants_raise_instead_of_falling_back():
    with pytest.raises(NotImplementedError):
        _build("h36m_zju/danbo_base.txt", extra=["--agg_type", "softmax"])
    with pytest.raises(NotImplementedError):
        _build("h36m_zju/danbo_base.txt", extra=["--gnn_backbone", "PNBGNN"])


def test_fused_training_step_reports_the_options_it_does_not_cover():
    """train_engine.supported -> None for the shipped DANBO configurations, and a reason (=> the trainer takes the autograd path,
    whose caster raises NotImplementedError) for every option the C step has no field for"""
    from core import train_engine
    args, (tr, *_r) = _build("perfcap/danbo_fast.txt")
    assert train_engine.supported(args, tr["ray_caster"]) is None
    for flag, value, reason in [("lindisp", True, "lindisp"), ("ray_noise_std", 0.5, "ray_noise_std"), ("loss_fn", "Huber", "loss_fn"),
                                ("weight_decay", 1e-4, "weight decay"), ("opt_pose", True, "opt_pose"), ("N_importance", 0, "sampling"),
                                ("density_type", "softplus", "density_type")]:
        a = copy.copy(args)
        setattr(a, flag, value)
        got = train_engine.supported(a, tr["ray_caster"])
        assert got is not None and reason in got, (flag, got)
    a_args, (a_tr, *_r) = _build("h36m_zju/anerf_base.txt")
    assert "network" in train_engine.supported(a_args, a_tr["ray_caster"])


@pytest.mark.parametrize("cfg_file,name", [("h36m_zju/danbo_base.txt", "danbo_base"), ("h36m_zju/anerf_base.txt", "anerf_base")])
def test_checkpoint_wire_format_equals_the_reference_manifest(cfg_file, name):
    """tests/golden/ckpt_manifest.json = top-level keys and name -> shape of `caster.state_dict()` of the REFERENCE
    (what trainer.py:597-618 saves): our caster must produce and accept exactly that dictionary"""
    import json
    manifest = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ckpt_manifest.json")))[name]
    args, (tr, te, *_rest) = _build(cfg_file)
    caster = te["ray_caster"]
    ck = caster.state_dict()
    assert set(ck) == set(manifest)
    for top, sub in manifest.items():
        ours = {k: list(v.shape) for k, v in ck[top].items()}
        assert ours == sub, (top, set(ours) ^ set(sub))
    # a checkpoint with the reference's layout (random tensors of the manifest's shapes) loads strictly
    one = {k: torch.randn(*shape) if shape else torch.tensor(1.5) for k, shape in manifest["network_fn_state_dict"].items()}
    fake = {top: dict(one) for top in manifest}       # single_net: the reference saves the same network twice
    caster.load_state_dict(fake)
    for k, v in caster.network.state_dict().items():
        assert torch.equal(v, fake["network_fn_state_dict"][k]), k


def test_args_txt_round_trip_matches_the_reference_reader(tmp_path):
    """tests/golden/args_txt.json: an args.txt as the reference trainer writes it (run_nerf.py:590-594) and the argv
    the reference's own txt_to_argstring makes of it"""
    import json
    from core.config import parse_args, txt_to_argstring
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "args_txt.json")))
    path = tmp_path / "args.txt"
    path.write_text(g["args_txt"])
    assert txt_to_argstring(str(path)) == g["argv"]
    assert txt_to_argstring(str(path), ignore_config=True) == g["argv_ignore_config"]
    args = parse_args(txt_to_argstring(str(path), ignore_config=True))
    assert (args.nerf_type, args.netwidth, args.N_samples, args.N_importance, args.voxel_res, args.agg_backbone,
            args.opt_framecode, args.use_volume_near_far, args.loss_fn) == ("danbo", 256, 96, 48, 16, "vox_MIXGNN", True, False, "L1")


def test_kp_to_valid_rays_matches_reference():
    """render-time ray selection: image box of the bounding cylinder, flat pixel indices and the rays
    of those pixels, against the reference's kp_to_valid_rays (tests/golden/valid_rays.npz)"""
    from core.utils.ray_utils import kp_to_valid_rays
    g = golden("valid_rays")
    rays, idxs, cyl, boxes = kp_to_valid_rays(torch.tensor(g["cams"]), int(g["H"]), int(g["W"]), float(g["focal"]),
                                              kps=torch.tensor(g["kps"]), ext_scale=0.001)
    assert max_err(cyl.numpy(), g["cyl"]) < 1e-6
    assert np.array_equal(np.array([[b[0], b[1]] for b in boxes]), g["boxes"])
    assert [len(i) for i in idxs] == list(g["n_valid"])
    assert np.array_equal(idxs[0].numpy(), g["idx0"]) and np.array_equal(idxs[3].numpy(), g["idx3"])
    assert max_err(rays[3][0].numpy(), g["rays_o3"]) < 1e-6 and max_err(rays[3][1].numpy(), g["rays_d3"]) < 1e-6
    tl, br = boxes[0]
    assert 0 < tl[0] < br[0] < int(g["W"]) - 1 and 0 < tl[1] < br[1] < int(g["H"]) - 1   # an interior box, not the frame


def test_linear16_isa_keeps_its_hands_off_the_in_flight_row_registers(tmp_path):
    """k_linear16 requests its input rows two k-steps ahead into fixed physical registers v[208:223] and keeps the weight fragments
    of two groups in v[224:255] (mlp16_core.hpp group_mfma), all of which only its inline asm names: the kernel is compiled with
    amdgpu_num_vgpr(208).  Nothing the compiler generates may touch them (a copy or spill of a register whose load is in flight
    reads stale data), every MFMA sits inside the asm, and the compiler must not add vmcnt waits of its own inside the k-step loop
    (they would drain the weight ring).  Checked on the gfx950 ISA of every instantiation."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "danbo-pytorch_amd", "csrc", "k_linear16.hip")
    out = str(tmp_path / "k_linear16.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out, src],
                   check=True, capture_output=True)
    text = open(out).read()
    pinned = r"(20[89]|21\d|22\d|23\d|24\d|25[0-5])"
    high = re.compile(r"\bv" + pinned + r"\b|v\[" + pinned + ":")
    # every instantiation <NH, NP, TRACE, FRAG>; FRAG = activations in fragment order (danbo_linear16_fwd_frag)
    names = re.findall(r"^(_ZN5danbo10k_linear16ILi(\d)ELi(\d)ELb(\d)ELi(\d+)EEEvNS_9Lin16ArgsE):", text, re.M)
    seen = set()
    for name, nh, np_, trace, frag in names:
        nh, np_, trace, frag = int(nh), int(np_), int(trace), int(frag)
        seen.add((nh, np_, trace, frag))
        body = text[text.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")].split("\n")
        assert not any("scratch_" in l for l in body), ("register spills", name)
        # outside the inline asm nothing names v208 .. v255 and there is no MFMA
        in_asm, foreign, mfma = False, [], 0
        for l in body:
            in_asm = True if "ASMSTART" in l else (False if "ASMEND" in l else in_asm)
            if not in_asm and (high.search(l) or "v_mfma" in l):
                foreign.append(l.strip())
            mfma += "v_mfma_f32_16x16x32_f16" in l
        assert not foreign, (name, foreign[:4])
        # two unrolled k-steps of NH chunks, 8 groups of 6 (NP = 6: the last chunk of a k-step has 6 groups); FRAG & 16 (the colour
        # epilogue of A-NeRF's head layer, round 6): + 15 tiles x 3 products in each of the two unrolled tile ends -- inline asm too
        color = bool(frag & 16)
        assert mfma == 2 * 6 * (8 * (nh - 1) + (np_ or 8)) + (2 * 45 if color else 0), (name, mfma)
        touching = [l.strip() for l in body if high.search(l)]
        loads = [l for l in touching if l.startswith("global_load_dwordx4 v[2")]
        takes = [l for l in touching if re.match(r"v_mov_b32 v\d+, v2(0[89]|1\d|2[0-3])$", l)]
        frags = [l for l in touching if l.startswith("ds_read_b128 v[2") or l.startswith("v_mfma")]
        # requests: three in the prologue + one per unrolled k-step, 2 loads each (a kernel with one part in rows and one in
        # fragment order carries both address forms); takes: 8 registers in the prologue and in each of the two k-steps
        mixed = frag in (1, 5, 6, 12, 14, 17)   # (12 / 14: the A-NeRF encoder table instead of rows, danbo_linear16_fwd_enc)
        assert len(loads) == (20 if mixed else 10) and len(takes) == 24 and len(touching) == len(loads) + len(takes) + len(frags), (name, touching)
        waits = sorted(re.search(r"vmcnt\(\d+\)", l).group(0) for l in body if "s_waitcnt" in l and "vmcnt" in l)
        # bias table, prologue, one per row-tile end in each of the two unrolled k-steps, final drain; one per hand-over
        if color:
            # the colour epilogue sits at the row-tile end, behind the vmcnt(0) that has drained ring and row loads: its own loads
            # (cutoff weights, camera index, the batched gathers) are waited for there with vmcnt(0); the k-step loop's hand-over
            # waits stay the counted ones
            assert waits.count("vmcnt(6)") == 2 * nh and set(waits) == {"vmcnt(0)", "vmcnt(6)"}, (name, waits)
        else:
            assert waits == sorted(["vmcnt(0)"] * 5 + ["vmcnt(6)"] * (2 * nh)), (name, waits)
    # what the launcher dispatches to
    shapes = {(1, 0), (1, 8), (2, 0), (2, 6), (2, 8)}
    want = ({(nh, np_, 0, fr) for nh, np_ in shapes for fr in (0, 4, 5, 6)} | {(1, 0, 0, 1), (1, 8, 0, 1), (2, 0, 1, 0)}
            | {(2, 6, 0, 12), (2, 6, 0, 14)} | {(1, 8, 0, 17), (1, 0, 0, 17)})
    assert seen == want, (seen ^ want)


def test_no_packed_fp32_instruction_takes_its_low_half_from_src1s_high_dword(tmp_path):
    """The gfx950 erratum of round 6 (csrc/common.hpp DANBO_NO_PK_F32; tools/probe/cview_probe.hip): v_pk_fma_f32 / v_pk_mul_f32 /
    v_pk_add_f32 with op_sel = [x, 1, ..] -- the low half of the result from src1's HIGH dword -- are wrong in lanes 48 .. 63 while
    another wavefront of the CU executes MFMAs.  The compiler emits the form at will; the kernels where it did are compiled without
    packed fp32 instructions.  This disassembles the device code of the BUILT library (every kernel that ships) and fails if any
    instruction of that form is left, so a kernel that acquires one is caught at build time, not as a non-repeatable training step."""
    import shutil
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    so = os.path.join(ROOT, "danbo-pytorch_amd", "libdanbo_hip.so")
    if not (os.path.exists(objdump) and os.path.exists(so)):
        pytest.skip("no llvm-objdump / library")
    work = tmp_path / "co"
    work.mkdir()
    lib = str(work / "lib.so")
    shutil.copy(so, lib)
    subprocess.run([objdump, "--offloading", lib], check=True, capture_output=True, cwd=str(work))
    objs = [str(work / f) for f in os.listdir(work) if "amdgcn" in f and f.endswith("gfx950")]
    assert len(objs) >= 15, os.listdir(work)
    packed = vulnerable = 0
    where = []
    kernel = None
    for o in objs:
        text = subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True).stdout
        for line in text.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
            if m:
                kernel = m.group(1)
            if re.search(r"v_pk_(fma|mul|add)_f32", line):
                packed += 1
                if re.search(r"op_sel:\[[01],1", line):
                    vulnerable += 1
                    where.append((kernel, line.split("//")[0].strip()))
    assert packed > 1000, packed              # the scan sees the packed instructions the library does use
    assert vulnerable == 0, where[:8]


def test_dw16_isa_leaves_the_producers_load_registers_alone(tmp_path):
    """k_dw16's producer wavefronts request their operands two steps ahead with inline-asm loads into the fixed accumulation
    registers a0 .. a95 and wait for them by count (csrc/k_dw16_regs.inc): nothing the compiler generates for the producer code may
    name those registers (a value parked there would be overwritten by a load in flight), the kernel must not spill, must fit two
    wavefronts per SIMD (256 registers), and its waits must be the counted ones."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "danbo-pytorch_amd", "csrc", "k_dw16.hip")
    out = str(tmp_path / "k_dw16.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out, src],
                   check=True, capture_output=True)
    text = open(out).read()
    meta = re.search(r"\.name:\s+_ZN5danbo6k_dw16ENS_6DwArgsE\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text)
    assert meta is not None and int(meta.group(1)) <= 256, meta and meta.group(1)
    body = text[text.index("_ZN5danbo6k_dw16ENS_6DwArgsE:"):]
    body = body[:body.index(".Lfunc_end")].split("\n")
    assert not any("scratch_" in l for l in body)
    pinned = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
    first_load = next(i for i, l in enumerate(body) if "global_load_dwordx4 a[" in l)
    in_asm, loads, foreign = False, 0, []
    for i, l in enumerate(body):
        in_asm = True if "ASMSTART" in l else (False if "ASMEND" in l else in_asm)
        code = l.split(";")[0]
        if in_asm:
            loads += "global_load_dwordx4 a[" in code
            continue
        if i < first_load:          # the consumers' code (MFMA accumulators live in a0 .. a127 there) is emitted first
            continue
        for m in pinned.finditer(code):
            lo = int(m.group(1) or m.group(2))
            if lo < 96:
                foreign.append(code.strip())
    assert not foreign, foreign[:4]
    assert loads > 0 and loads % 12 == 0, loads                                     # 12 loads per lane and step, always
    waits = {re.search(r"vmcnt\(\d+\)", l).group(0) for l in body[first_load:] if "s_waitcnt" in l and "vmcnt" in l}
    assert "vmcnt(12)" in waits, waits


def test_ring_kernels_do_not_spill(tmp_path):
    """The kernels that stream weights through an LDS ring with hand-counted vmcnt waits must not spill: a scratch access is a
    VMEM operation the compiler waits for with vmcnt(0), which drains the ring (k_pe_mlp16 with 196 B of spills moved 10x the HBM
    traffic).  Checked on the gfx950 ISA."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    for src, kernels in (("k_mlp16.hip", ["k_pe_mlp16"]), ("k_assign16.hip", ["k_assign16ILb0E", "k_assign16ILb1E"])):
        out = str(tmp_path / (src + ".s"))
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out,
                        os.path.join(ROOT, "danbo-pytorch_amd", "csrc", src)], check=True, capture_output=True)
        text = open(out).read()
        for k in kernels:
            meta = re.search(r"\.name:\s+\S*" + k + r"\S*\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)
            assert meta is not None, k
            assert int(meta.group(1)) == 0, (k, meta.group(1))
            body = text[text.index(re.search(r"^(_ZN5danbo\S*" + k + r"\S*):", text, re.M).group(1) + ":"):]
            body = body[:body.index(".Lfunc_end")]
            assert "scratch_" not in body and "buffer_store" not in body, k
    # The training trunk's kernels run the same ring with stores in between.  They may spill a few tile-level values (pointers of
    # the row outputs) -- but no scratch access between the first and the last MFMA of the forward's layer loop, and in the
    # backward only in the once-per-tile staging branch.
    for src, k, max_in_loop in (("k_mlp16.hip", "k_train_mlp_fwd", 0), ("k_mlp16_bwd.hip", "k_train_mlp_bwd", 0)):
        out = str(tmp_path / (src + ".t.s"))
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out,
                        os.path.join(ROOT, "danbo-pytorch_amd", "csrc", src)], check=True, capture_output=True)
        text = open(out).read()
        body = text[text.index(re.search(r"^(_ZN5danbo\S*" + k + r"\S*):", text, re.M).group(1) + ":"):]
        body = body[:body.index(".Lfunc_end")].split("\n")
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        inside = [l for i, l in enumerate(body) if "scratch_" in l and mf[0] < i < mf[-1]]
        assert len(inside) <= max_in_loop, (k, inside)
    # Round 5: K3 and the training forward keep the weight fragments of chunk_mfma2 in the PINNED registers v224..v255 across
    # compiler-generated code (mlp16_core.hpp; the kernels are compiled with amdgpu_num_vgpr(224)); the training backward too.  No instruction outside the
    # inline asm may name one of them, the kernels use no scratch at all, and every MFMA of theirs is inside the asm.
    high = re.compile(r"\bv(22[4-9]|2[34]\d|25[0-5])\b|v\[(\d+):(\d+)\]")
    for k, n_mfma, asm_file in (("k_pe_mlp16", 960, "k_mlp16.hip.s"), ("k_train_mlp_fwd", 960, "k_mlp16.hip.s"),
                                ("k_train_mlp_bwd", 912, "k_mlp16_bwd.hip.t.s")):      # 20 chunk sites x 48; 4 + 8 x 48 + 8 x 42
        text = open(str(tmp_path / asm_file)).read()
        meta = re.search(r"\.name:\s+\S*" + k + r"\S*\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)
        assert int(meta.group(1)) == 0, (k, meta.group(1))
        body = text[text.index(re.search(r"^(_ZN5danbo\S*" + k + r"\S*):", text, re.M).group(1) + ":"):]
        body = body[:body.index(".Lfunc_end")].split("\n")
        in_asm, foreign, mfma_outside, mfma_inside = False, [], 0, 0
        for l in body:
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            elif in_asm:
                mfma_inside += "v_mfma" in l
            else:
                code = l.split(";")[0]
                mfma_outside += "v_mfma" in code
                for m in high.finditer(code):
                    if m.group(1) or (m.group(3) and int(m.group(3)) >= 224):
                        foreign.append(code.strip())
        assert not foreign, (k, foreign[:4])
        assert mfma_outside == 0 and mfma_inside == n_mfma, (k, mfma_outside, mfma_inside)


def test_k3_32x32_form_register_discipline(tmp_path):
    """k_pe_mlp32 (K3 in the 32x32x16 form) names its own registers in inline assembly: the two AccVGPR result banks, the fragment
    double buffers, the epilogue temporaries (v184 .. v255).  The compiler only honours them ACROSS one asm statement; what keeps
    them safe between statements is that it has no reason to take them.  Checked on the gfx950 ISA: between the first and the last
    MFMA no compiler-generated instruction names v184+ or any AccVGPR (this compiler DID take a[0:59] for the colour head's weights
    behind the last MFMA -- why the view layer's accumulators are asm operands, not pinned), none touches M0 (every LDS-DMA load
    is an asm statement with its own M0 write: with builtin loads the compiler hoisted its M0 initialisation across the groups and
    the staging loads went into the ring), there is no scratch, and every MFMA sits inside the asm."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = str(tmp_path / "k_mlp32.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", out,
                    os.path.join(ROOT, "danbo-pytorch_amd", "csrc", "k_mlp32.hip")], check=True, capture_output=True)
    text = open(out).read()
    k = "k_pe_mlp32"
    meta = re.search(r"\.name:\s+\S*" + k + r"\S*\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)
    assert meta is not None and int(meta.group(1)) == 0, meta and meta.group(1)
    body = text[text.index(re.search(r"^(_ZN5danbo\S*" + k + r"\S*):", text, re.M).group(1) + ":"):]
    body = body[:body.index(".Lfunc_end")].split("\n")
    assert not any("scratch_" in l for l in body)
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    # 2 x 13 k-substeps of the encoding x 24 | three dense-layer sites of 16 x 24 (+ the skip layer's second form of k-substep 0) | view 16 x 12
    assert len(mf) == 2 * 13 * 24 + 3 * 16 * 24 + 24 + 16 * 12, len(mf)
    vreg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
    areg = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
    in_asm, foreign, mfma_outside, m0_outside = False, [], 0, []
    for i, l in enumerate(body):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif not in_asm:
            code = l.split(";")[0]
            mfma_outside += "v_mfma" in code
            if re.search(r"\bm0\b", code):
                m0_outside.append(code.strip())
            if mf[0] < i < mf[-1]:
                if areg.search(code):
                    foreign.append(code.strip())
                for m in vreg.finditer(code):
                    if int(m.group(1) or m.group(3)) >= 184:
                        foreign.append(code.strip())
    assert mfma_outside == 0
    assert not foreign, foreign[:4]
    assert not m0_outside, m0_outside[:4]


def test_fragment_order_buffer_layout_on_cpu():
    """hip_ops.FragBuffer (the activation layout between the layers of a k_linear16 trunk, include/danbo_hip.h
    danbo_linear16_fwd_frag): element (16 g + n, 32 s + 16 h + 4 q + i) lives at [g][s][h][q][n][i], rows padded to the 128-row
    tile -- pure index arithmetic, checked without a GPU; and the packing permutation a consumer layer uses for such an input:
    k-slot 8 q + e of k-step s carries column 32 s + 16 (e / 4) + 4 q + e % 4, i.e. lane (n, q) finds its 8 k-slots of k-step s
    in the two 16-byte pieces [g][s][0][q][n][:] and [g][s][1][q][n][:]"""
    from core import hip_ops as ops
    rng = np.random.default_rng(5)
    M, C = 300, 96
    x = torch.from_numpy(rng.normal(size=(M, C)).astype(np.float32))
    fb = ops.FragBuffer.from_rows(x)
    assert fb.data.numel() == 384 * C and torch.equal(fb.rows(), x)
    d = fb.data.view(-1, C // 32, 2, 4, 16, 4)
    for g, s, h, q, n, i in ((0, 0, 0, 0, 0, 0), (2, 1, 1, 3, 5, 2), (18, 2, 0, 1, 11, 3)):
        assert float(d[g, s, h, q, n, i]) == float(x[16 * g + n, 32 * s + 16 * h + 4 * q + i])
    assert float(d[18, 0, 0, 0, 12, 0]) == 0.0                         # row 300: padding
    # the consumer's view: lane (n, q), k-slot 8 q + e  <->  column 32 s + 16 (e // 4) + 4 q + e % 4
    g, s, n, q = 3, 2, 7, 2
    lane_vals = torch.cat([d[g, s, 0, q, n], d[g, s, 1, q, n]])
    want = torch.stack([x[16 * g + n, 32 * s + 16 * (e // 4) + 4 * q + e % 4] for e in range(8)])
    assert torch.equal(lane_vals, want)
    with pytest.raises(ValueError):
        ops.FragBuffer(10, 100, "cpu")


def test_assignment_visualisations_match_the_reference():
    """get_confidence_rgb / get_entropy_rgb (NeRF.raw2outputs(render_confd / render_entropy), reference misc.py:620-673) against the
    reference's outputs on random logits (tests/golden/confd_colours.npz)"""
    from core.networks.misc import get_confidence_rgb, get_entropy_rgb
    g = golden("confd_colours")
    c = torch.tensor(g["confd"])
    assert np.array_equal(get_confidence_rgb(c).numpy(), g["confidence_rgb"])
    assert max_err(get_entropy_rgb(c).numpy(), g["entropy_rgb"]) < 1e-6


def test_custom_operators_register_and_propagate_shapes_without_a_gpu():
    """core/custom_ops.py: every torch.ops.danbo.* operator is registered with a schema and a fake kernel (shape propagation on the
    meta device works without the HIP library being called), and refuses CPU tensors instead of falling back to anything."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "danbo-pytorch_amd"))
    from core import custom_ops  # noqa: F401
    M = lambda *s, dt=torch.float32: torch.empty(*s, device="meta", dtype=dt)  # noqa: E731
    R, S, G, n = 6, 5, 2, 11
    # composite
    out = torch.ops.danbo.composite(M(R, S, 4), M(R, S), M(R, 3), 1.0, None)
    assert [tuple(o.shape) for o in out] == [(R, 3), (R,), (R,), (R, S), (R, S)]
    # bone gather
    pf = torch.ops.danbo.bone_gather(M(G, 24, 240), M(24, 3), M(R, S, 3), M(G, 24, 4, 4), M(24, 4, 4), M(n, dt=torch.int32))
    assert tuple(pf.shape) == (n, 24, 15)
    # assign + blend
    ap = [M(24, 15, 32), M(1, 24, 24), M(1, 24, 24), M(32), M(24, 32, 32), M(24, 1, 32), M(24, 32, 1), M(24, 1, 1)]
    h, p, confd = torch.ops.danbo.assign_blend(M(G, 24, 240), M(24, 3), M(R, S, 3), M(G, 24, 4, 4), M(24, 4, 4), M(n, dt=torch.int32),
                                               M(R * S, dt=torch.int32), ap)
    assert tuple(h.shape) == (n, 16) and tuple(p.shape) == (n, 24) and tuple(confd.shape) == (n, 24)
    # PE + MLP
    mp = ([M(256, 195)] + [M(256, 256)] * 4 + [M(256, 451)] + [M(256, 256)] * 2 + [M(256)] * 8
          + [M(1, 256), M(1), M(256, 256), M(256), M(128, 256 + 155), M(128), M(3, 128), M(3)])
    raw = torch.ops.danbo.pe_mlp(M(n, 15), M(n, dt=torch.int32), M(R, 155), mp)
    assert tuple(raw.shape) == (n, 4)
    # pose -> volumes
    gp = [M(24, 66, 128), M(1, 24, 24), M(1, 24, 24), M(128), M(24, 128, 128), M(1, 24, 24), M(1, 24, 24), M(128), M(24, 128, 128), M(1, 24, 128),
          M(24, 128, 240), M(1, 24, 240)]
    vol, scratch = torch.ops.danbo.pose_volumes(M(G, 24, 3), 5, gp)
    assert tuple(vol.shape) == (G, 24, 240) and scratch.numel() == 3 * G * 24 * 128
    with pytest.raises(RuntimeError):
        torch.ops.danbo.pose_volumes(torch.zeros(G, 24, 3), 5, [torch.zeros(1)] * 12)
    # A-NeRF chain (forward only)
    anp = ([M(448, 432)] + [M(448, 448)] * 4 + [M(448, 880)] + [M(448, 448)] * 2 + [M(448)] * 8
           + [M(1, 448), M(1), M(448, 448), M(448), M(224, 448 + 648 + 128), M(224), M(3, 224), M(3), M(24), M(24), M(20, 128)])
    raw = torch.ops.danbo.anerf_cutoff_pe_mlp(M(R, S, 3), M(R, 3), M(G, 24, 4, 4), M(24, 4, 4), M(R, dt=torch.int64), anp, 20.0, 7, 4)
    assert tuple(raw.shape) == (R, S, 4)
    # no CPU fallback
    with pytest.raises(RuntimeError):
        torch.ops.danbo.pe_mlp(torch.zeros(n, 15), torch.zeros(n, dtype=torch.int32), torch.zeros(R, 155), [torch.zeros(1)] * 24)
    with pytest.raises(RuntimeError):
        torch.ops.danbo.assign_blend(torch.zeros(G, 24, 240), torch.zeros(24, 3), torch.zeros(R, S, 3), torch.zeros(G, 24, 4, 4), torch.zeros(24, 4, 4),
                                     torch.zeros(n, dtype=torch.int32), torch.zeros(R * S, dtype=torch.int32), [torch.zeros(1)] * 8)


def test_fused_engine_adam_state_resumes_under_plain_adam(tmp_path):
    """The fused engine's Adam state (train_engine.adopt_adam_state: exp_avg / exp_avg_sq as views of flat buffers, one DISTINCT
    `step` tensor per parameter) must survive state_dict -> torch.save -> torch.load -> load_state_dict into a plain torch.optim.Adam
    -- the reference's loader (core/raycasters.py:63-86), and this repo's autograd path -- with `step` advancing by ONE per
    optimizer.step().  (A shared step tensor stays shared through the round trip and advances by the parameter count: ADVICE r3.)"""
    from core.train_engine import adopt_adam_state
    torch.manual_seed(0)
    shapes = [(5, 3), (7,), (2, 2), (4,), (3, 3)]
    params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    opt = torch.optim.Adam(params, lr=1e-2)
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += p.numel()
    flat_m, flat_v = torch.zeros(off), torch.zeros(off)
    steps = adopt_adam_state(opt, params, offsets, flat_m, flat_v)
    assert len({id(s) for s in steps}) == len(params) and len({s.data_ptr() for s in steps}) == len(params)
    t = 8
    for _ in range(t):                                   # what DanboTrainEngine.adam_step does to the counters
        torch._foreach_add_(steps, 1.0)
    flat_m.uniform_(-1, 1)
    flat_v.uniform_(0.1, 1)
    path = tmp_path / "opt.tar"
    torch.save({"optimizer_state_dict": opt.state_dict()}, path)
    params2 = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt2 = torch.optim.Adam(params2, lr=1e-2)
    opt2.load_state_dict(torch.load(path, weights_only=False)["optimizer_state_dict"])
    for p in params2:
        p.grad = torch.ones_like(p)
    opt2.step()
    for p in params2:
        assert float(opt2.state[p]["step"]) == t + 1
    # and the engine adopts a loaded state: moments copied into the flat buffers, the count taken over
    flat_m2, flat_v2 = torch.zeros(off), torch.zeros(off)
    steps2 = adopt_adam_state(opt2, params2, offsets, flat_m2, flat_v2)
    assert all(float(s) == t + 1 for s in steps2) and len({s.data_ptr() for s in steps2}) == len(params)
    assert torch.equal(flat_m2[offsets[1]:offsets[1] + 7], opt2.state[params2[1]]["exp_avg"].reshape(-1))
    assert flat_m2.abs().sum() > 0 and opt2.state[params2[0]]["exp_avg"].data_ptr() == flat_m2.data_ptr()


def test_philox_restatement_known_answers():
    """the numpy Philox4x32-10 the GPU draw kernel is checked against (tests/helpers.py), on the Random123 known-answer vectors"""
    from helpers import philox4x32_10
    kat = [([0, 0, 0, 0], (0, 0), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, (0xffffffff, 0xffffffff), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0), [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert [int(x) for x in philox4x32_10([ctr], key)[0]] == want


def test_row_spans_of_strided_tensors():
    """hip_ops.row_span (what danbo_gather_rows is told about a tensor): strided rows stay views, 8-byte elements count as word
    pairs, broadcast rows have stride 0, anything whose rows are not contiguous inside is copied first; the ctypes mirror of
    DanboRowSpan has the C struct's size"""
    import ctypes
    import torch
    from core import _hip, hip_ops
    assert ctypes.sizeof(_hip.DanboRowSpan) == 32 and _hip.MAX_ROW_SPANS == 12
    a = torch.arange(3072 * 24 * 16, dtype=torch.float32).reshape(3072, 24, 4, 4)
    t, rows, words, stride = hip_ops.row_span(a[::192])
    assert t.data_ptr() == a.data_ptr() and (rows, words, stride) == (16, 384, 192 * 384)
    d = torch.arange(10, dtype=torch.int64)
    assert hip_ops.row_span(d)[1:] == (10, 2, 2) and hip_ops.row_span(d[::3])[1:] == (4, 2, 6)
    e = torch.ones(1, 3).expand(7, 3)
    assert hip_ops.row_span(e)[1:] == (7, 3, 0)
    f = torch.arange(50 * 7, dtype=torch.float32).reshape(50, 7).t()
    t, rows, words, stride = hip_ops.row_span(f)
    assert t.is_contiguous() and (rows, words, stride) == (7, 50, 50) and torch.equal(t, f)
    assert hip_ops.row_span(torch.tensor(3.0))[1:] == (1, 1, 1)
