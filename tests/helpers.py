"""Shared test plumbing: fixtures on disk, seeded worlds, oracle adapters."""
import hashlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def sha(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()[:16]


def weights_checksum(net):
    return sha(*[p for _, p in sorted(net.state_dict().items()) if p.dtype == torch.float32])


def oracle_table(tbl):
    """SyntheticSMPL -> dict of CPU tensors in the layout oracle.smpl_forward takes."""
    import anim_nerf_amd as ana
    bm = ana.SMPL(data_struct=tbl)
    return dict(v_template=bm.v_template, shapedirs=bm.shapedirs, posedirs=bm.posedirs,
                J_regressor=bm.J_regressor, parents=bm.parents, lbs_weights=bm.lbs_weights,
                extra_joints_idxs=bm.vertex_joint_selector.extra_joints_idxs)


def tdict(d, prefix="pose_"):
    names = ("betas", "global_orient", "body_pose", "transl")
    return {n: torch.from_numpy(d[prefix + n]) for n in names}


def net_params(net):
    """CPU copies of a NeRF module's parameters under the reference's state-dict keys."""
    return {k: v.detach().cpu() for k, v in net.named_parameters()}


def seeded_model(tbl, seed, use_unpose, gain=1.0, shift=(0.0, 0.0), device=None, **kw):
    """Our AnimNeRF with the same seeded init (and the same sigma gain) as a render fixture."""
    import anim_nerf_amd as ana
    torch.manual_seed(int(seed))
    m = ana.AnimNeRF(body_model_table=tbl, freqs_dir=0, use_view=False, use_unpose=bool(use_unpose), use_fine=True, **kw)
    if float(gain) != 1.0:
        sh = np.atleast_1d(shift)
        with torch.no_grad():
            for net, s in ((m.nerf, sh[0]), (m.nerf_fine, sh[1])):
                net.sigma.weight.mul_(float(gain))
                net.sigma.bias.mul_(float(gain)).add_(float(s))
    m.eval()
    return m.to(device) if device is not None else m


def rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor)."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs() / b.abs().clamp_min(floor)).max().item()
