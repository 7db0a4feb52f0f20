"""CPU tests of host-side logic: C-ABI surface, MLP pack/dataflow algebra, module surface, sharding."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import mlp_emulator as emu
from helpers import net_params, seeded_model
from oracle import animnerf_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """Every function include/animnerf_hip.h declares is exported by the built library and bound in _lib."""
    import anim_nerf_amd as ana
    hdr = open(os.path.join(ROOT, "include", "animnerf_hip.h")).read()
    declared = set(re.findall(r"\b(anr_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("anr_mlp_params")
    assert declared == set(ana._lib.SIGNATURES), declared ^ set(ana._lib.SIGNATURES)
    lib = ana._lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.anr_version() == 100
    assert lib.anr_mlp_pack_bytes(1) == 10240 + 1176 * 1024 and lib.anr_mlp_pack_bytes(0) == 10240 + 2352 * 1024


def test_bad_arguments_are_reported_not_crashed():
    import anim_nerf_amd as ana
    lib = ana._lib.load()
    rc = lib.anr_composite(None, None, None, 8, None, 4, 64, 1, None, None, None, None, None)
    assert rc == -1 and b"null pointer" in lib.anr_last_error()
    # every entry point validates before it touches the device: null pointers / bad sizes come back as codes + text
    calls = {
        "anr_ray_gen": None, "anr_compact_valid": (None, 8, None, None, None, 0, None),
        "anr_mlp_forward": (None, 1, None, 8, None, None),
        "anr_mlp_forward_indexed": (None, 1, None, None, None, 8, None, None),
        "anr_mlp_forward_rays": (None, 1, None, 8, None, 4, 8, None, None),
        "anr_mlp_forward_save": (None, 1, None, 8, None, None, None),
        "anr_mlp_backward": (None, 1, None, None, None, 8, None),
        "anr_encode": (None, 4, 8, 0, None, None), "anr_encode_backward": (None, 4, None, 8, None, None),
        "anr_composite_masked": (None, None, None, 8, None, None, 4, 64, 1, None, None, None, None, None),
        "anr_sample_fine_merge": (None, None, None, 0, 4, 64, 64, None, None, None, None),
        "anr_sample_fine_merge_u8": (None, None, None, 0, 4, 64, 64, None, None, None, None),
        "anr_knn": (None, None, 1, 6890, 8, None, None, None),
        "anr_warp_points_lean": (None, 0, None, 8, None, 4, None, None, None, 1, 6890, 24, 8, 0.2, 1, None, None, None, None,
                                 None, None, None, None, None, None, None, None, None, 0, None),
        # round 2: the training-step kernels
        "anr_knn_k": (None, None, 3, 1, 6890, 8, 9, None, None, None),
        "anr_compact_ordered": (None, 8, None, None, None, None, None, None),
        "anr_expand_rows": (None, None, 8, 3, 0.0, None, None),
        "anr_mlp_head_grad": (None, None, None, None, 8, 64, 0, None, None),
        "anr_tangent_quads": (None, 8, 16, None, None),
        "anr_sample_coarse_backward": (None, None, None, 8, 64, None, None),
        "anr_merge_backward": (None, None, 8, 96, 64, None, None),
        "anr_mlp_wgrad": (1, None, None, None, None, 64, None, None, None),
        "anr_mlp_denc": (1, None, None, None, 64, None, None),
        "anr_mlp_backward_feature": (None, 1, None, None, None, None, 64, None),
        "anr_train_loss": (None, None, None, None),
        "anr_train_loss_backward": (None, None, None, None),
        "anr_frame_backward": (None,) * 3 + (1,) + (None,) * 6 + (6890, None, 1, None, 8, 0, None, None, None, None, None),
        "anr_frame_backward_adjoint": (None,) * 3 + (1,) + (None,) * 6 + (6890, None, 1, None, 8, 0, None, None, None, None, None),
        "anr_to_root_frame": (None, None, None, None, 1, 6890, 24, None, None, None, None, None, None),
        "anr_composite_sample": (None,) * 4 + (8, None, None, 0, 4, 64, 64, 1) + (None,) * 8,
        # round 3: device-side row counts, the fused view-dependent head, marching cubes, Adam
        "anr_mlp_forward_save_indexed": (None, 1, None, None, None, 8, None, None, None),
        "anr_mlp_backward_counted": (None, 1, None, None, None, 64, None, None),
        "anr_mlp_wgrad_counted": (1, None, None, None, None, 64, None, None, None, None),
        "anr_mlp_denc_counted": (1, None, None, None, 64, None, None, None),
        "anr_encode64_counted": (None, 4, 64, None, 0, None, None),
        "anr_encode_backward_counted": (None, 4, None, 64, None, None, None),
        "anr_mlp_head_grad_counted": (None, None, None, None, None, 64, 0, None, None),
        "anr_mlp_pack_view": (None, 1, 27, None, None),
        "anr_mlp_forward_view": (None, 1, None, None, 3, None, None, 8, None, None),
        "anr_mc_classify": (None, 8, 8, 8, 0.0, None, None, None, None, None),
        "anr_mc_emit": (None, 8, 8, 8, 0.0) + (None,) * 8,
        "anr_adam_step": (None, 1, None, None, 1, 0.9, 0.999, 1e-8, None),
        "anr_warp_points_reuse": (None, 0, None, 8, None, 4, None, None, None, 1, 6890, 24, 8, 0.2, 1) + (None,) * 13 + (0, None, None, None),
        "anr_warp_points_cells": (None, 0, None, 8, None, 4, None, None, None, 1, 6890, 24, 8, 0.2, 1) + (None,) * 13 + (0, None, None, None, 0, None),
        # round 6: the one-pass ray-march kernel, the frame set-up with the pose tables' row count
        "anr_ray_march": (None, None, 1, None, 8, 4, None, 64, None, 64, 1) + (None,) * 7,
        "anr_ray_march_warp": (None, None, 1, None, 8, 1, 4, None, 64, None, 64, 1, None, None, None, 6890, 24, 0.2) + (None,) * 7,
        "anr_frame_setup_rows": (None, 4) + (None, 1) + (None,) * 3 + (1,) + (None,) * 7 + (6890, 24, 10) + (None,) * 3 + (1, None, 8, 0) + (None,) * 15,
    }
    for name, args in calls.items():
        if args is None:
            continue
        assert len(args) == len(ana._lib.SIGNATURES[name][1]), name
        rc = getattr(lib, name)(*args)
        assert rc < 0, name
        assert len(lib.anr_last_error()) > 0, name
    assert lib.anr_mlp_pack_bytes(7) < 0 and lib.anr_mlp_bwd_pack_bytes(7) < 0 and lib.anr_warp_ws_ints(0, 5) < 0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ana.ops.points_from_rays(torch.zeros(1, 4, 8), torch.zeros(1, 4, 8))


@pytest.mark.parametrize("epf", [4, 8])
def test_mlp_dataflow_algebra(smpl_table, epf):
    """The pack layout + MFMA lane maps + slot algebra reproduce the reference MLP (float64 emulation)."""
    m = seeded_model(smpl_table, 7, True)
    P = {k: v.double().numpy() for k, v in net_params(m.nerf).items()}
    xyz = torch.rand(32, 3, generator=torch.Generator().manual_seed(3)).double() * 2 - 1
    rgb, sig = emu.run(P, xyz.numpy(), epf)
    P64 = {k: torch.from_numpy(v) for k, v in P.items()}
    rgb_o, sig_o = orc.mlp_forward(P64, xyz)
    np.testing.assert_allclose(rgb, rgb_o.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(sig, sig_o[:, 0].numpy(), rtol=1e-9, atol=1e-12)


def test_module_surface_matches_reference(smpl_table):
    """Constructor kwargs, state-dict keys and attributes callers touch."""
    import anim_nerf_amd as ana
    m = ana.AnimNeRF(body_model_table=smpl_table, freqs_dir=0, use_unpose=True, use_knn=True, use_fine=True,
                     dis_threshold=0.2, pose_dim=69)
    keys = set(m.state_dict().keys())
    for net in ("nerf", "nerf_fine"):
        for i in range(1, 9):
            assert f"{net}.xyz_encoding_{i}.0.weight" in keys and f"{net}.xyz_encoding_{i}.0.bias" in keys
        for k in ("xyz_encoding_final.weight", "dir_encoding.0.weight", "sigma.weight", "rgb.0.weight"):
            assert f"{net}.{k}" in keys
    for k in ("betas", "global_orient", "body_pose", "transl", "shapedirs", "faces_tensor", "v_template", "J_regressor",
              "posedirs", "parents", "lbs_weights", "vertex_joint_selector.extra_joints_idxs"):
        assert "body_model." + k in keys
    assert m.state_dict()["body_model.posedirs"].shape == (207, 20670)
    assert m.state_dict()["nerf.xyz_encoding_5.0.weight"].shape == (256, 319)
    for attr in ("set_latent_code", "set_body_model", "convert_to_body_model_space", "clac_ober2cano_transform",
                 "forward", "query_canonical_space", "unpose", "body_model", "dis_threshold", "use_fine", "nerf", "nerf_fine"):
        assert hasattr(m, attr)
    vr = ana.VolumeRenderer(n_coarse=64, n_fine=32, n_fine_depth=0, share_fine=False, white_bkgd=True)
    assert vr.noise_std == 1.0 and vr.lindisp


def test_smpl_host_matches_oracle(smpl_table):
    """body_model.SMPL (the product's per-frame host code) == oracle.smpl_forward on CPU tensors."""
    import anim_nerf_amd as ana
    from anim_nerf_amd import synthetic as syn
    from helpers import oracle_table
    bm = ana.SMPL(data_struct=smpl_table)
    pose = {k: torch.from_numpy(v) for k, v in syn.animated_pose_params(seed=5, bs=3).items()}
    o = bm(**pose)
    ref = orc.smpl_forward(oracle_table(smpl_table), **pose)
    for k in ("vertices", "joints", "joints_transform", "vertices_transform", "shape_offsets", "pose_offsets"):
        torch.testing.assert_close(o[k], ref[k], rtol=1e-5, atol=2e-6)


def test_shard_range_partitions_exactly():
    from anim_nerf_amd import shard_range
    for n in (0, 1, 7, 1024 * 1024, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_forward_mode_normals_equal_double_backward():
    """The forward-mode formulation of NeRF.get_normal (three tangents per point through the trunk, ReLU gates of the
    point, plain linear backward: what autograd.NormalFunction runs in the fused kernels, restated with tensor ops in
    tests/normal_reference.py) against autograd of autograd over the same layers (models/nerf.py:177-190), in fp64 on the
    CPU: values and all 18 weight gradients."""
    import anim_nerf_amd as ana
    from normal_reference import NormalFunctionTorch
    torch.manual_seed(0)
    net = ana.NeRF(freqs_dir=0, use_view=False).double()
    xyz = torch.rand(1, 200, 3, dtype=torch.double) * 1.2 - 0.6
    with torch.no_grad():                                   # sigma must straddle 0 or every normal is 0
        med = net._sigma_dense(xyz).median()
        net.sigma.weight.mul_(300)
        net.sigma.bias.mul_(300).add_(-300 * med)
    named = dict(net.named_parameters())

    def forward_mode(x):
        return NormalFunctionTorch.apply(x.reshape(-1, 3), 0.02, *[named[k] for k in NormalFunctionTorch.KEYS]).view(*x.shape)
    grads = []
    for fn in (forward_mode, net._normal_autograd):
        net.zero_grad()
        n = fn(xyz)
        (n ** 2).sum().backward()
        grads.append((n.detach(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    (n1, g1), (n2, g2) = grads
    assert 0.2 < (n2.abs().sum(-1) > 0).double().mean() < 0.8
    assert (n1 - n2).abs().max() < 1e-12 * n2.abs().max()
    assert set(g1) == set(g2) and len(g1) == 18
    for k in g2:
        assert (g1[k] - g2[k]).norm() <= 1e-12 * g2[k].norm(), k


def test_build_keeps_slp_vectoriser_off():
    """Every source is compiled with -fno-slp-vectorize (anim-nerf_amd/build.py: NO_SLP): the packed fp32 adds the SLP vectoriser
    emits made 1-2 % of replayed training steps differ from each other on the GPU.  The GPU-side regression test is
    test_replays_of_one_step_reproduce_its_gradients; this one keeps the flag from being dropped by an edit of the recipe."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "anim-nerf_amd", "build.py")
    src = open(here).read()
    assert 'NO_SLP = [] if os.environ.get("ANR_BUILD_SLP") else ["-fno-slp-vectorize"]' in src and "flags += NO_SLP" in src
    if not os.environ.get("ANR_BUILD_SLP"):
        spec = importlib.util.spec_from_file_location("_anr_build_recipe", here)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert mod.NO_SLP == ["-fno-slp-vectorize"]


def test_linked_library_holds_no_swapped_or_half_negated_packed_fp32():
    """The gate behind the flag: the linked library's gfx950 code objects are disassembled and no
    v_pk_add/mul/fma_f32 may swap an operand's halves (op_sel) or negate one half only — the forms the SLP vectoriser produced
    and that returned a wrong lane next to other kernels.  build() runs the same scan after linking and rejects
    the library; here it is run on the library the tests load, whatever built it.  The packed forms that ARE there are the
    neighbour search's explicit float2 subtractions (both halves negated alike, in every kernel that inlines warp_core.h's
    scan_cluster), listed so that a new kind shows up."""
    import importlib.util
    import os
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_anr_build_recipe2", os.path.join(root, "anim-nerf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not (os.path.exists(mod.LIB_PATH) and os.path.exists(os.path.join(mod._LLVM_BIN, "llvm-objdump"))):
        pytest.skip("no built library / no llvm-objdump here")
    forms = mod.packed_fp32_forms()
    assert not [f for f in forms if f[2]], [f for f in forms if f[2]][:5]
    kernels = {k for k, _, _ in forms}
    # (the search's scan_cluster, inlined: the warp / KNN kernels and the one-pass kernel with the warp)
    assert all("warp" in k or "knn" in k or "ray_march_kernelILi0ELb1" in k or "ray_march_kernelILi2ELb1" in k for k in kernels), sorted(kernels)


def test_build_recipe_tracks_each_source_headers():
    """anim-nerf_amd/build.py recompiles an object when its source, a header it includes (transitively) or its command line
    changed: the dependency scan must see the headers the shared routines moved into this round (composite_core.h, warp_core.h)
    behind the sources that include them, and the public header behind everything."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_anr_build_recipe3", os.path.join(ROOT, "anim-nerf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    names = lambda src: {os.path.basename(p) for p in mod._deps(src)}
    assert {"ray_march.hip", "mlp_core.h", "warp_core.h", "composite_core.h", "anr_common.h", "animnerf_hip.h"} <= names("ray_march.hip")
    assert {"warp.hip", "warp_core.h", "anr_common.h", "animnerf_hip.h"} <= names("warp.hip") and "mlp_core.h" not in names("warp.hip")
    assert {"composite.hip", "composite_core.h"} <= names("composite.hip")
    assert all(h in mod.HEADERS for h in ("composite_core.h", "warp_core.h", "mlp_core.h"))
    assert set(mod.SOURCES) == {f for f in os.listdir(mod.CSRC) if f.endswith(".hip")}, "every .hip source is in the recipe"

