"""SMPL body model host module (per-frame work, a2 in SURVEY.md section 8).

Mirrors the interface of the reference's patched smplx (`smplx.create`, `SMPL.forward`,
smplx/body_models.py:44-387, smplx/lbs.py:152-404): same constructor meaning, same
buffer/parameter names (checkpoint compatible), same extra outputs
(`joints_transform`, `vertices_transform`, `shape_offsets`, `pose_offsets`).

It runs once or twice per frame on a 6890-vertex table (about 7 M MACs).  On the GPU without autograd it is three
HIP launches (`anr_smpl_forward`, csrc/smpl.hip); when gradients must reach the SMPL parameters (pose refinement) or
on the CPU it is the tensor-op form below, which torch autograd differentiates.
"""
from __future__ import annotations

import os
import pickle
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

# vertex ids of the 21 extra "joints" (reference smplx/vertex_ids.py:24-46 order used by
# smplx/vertex_joint_selector.py:33-67: face, feet, left finger tips, right finger tips)
_EXTRA_JOINT_VERTS = [332, 6260, 2800, 4071, 583,
                      3216, 3226, 3387, 6617, 6624, 6787,
                      2746, 2319, 2445, 2556, 2673,
                      6191, 5782, 5905, 6016, 6133]


class _JointSelector(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("extra_joints_idxs", torch.tensor(_EXTRA_JOINT_VERTS, dtype=torch.long))

    def forward(self, vertices, joints):
        return torch.cat([joints, vertices.index_select(1, self.extra_joints_idxs)], dim=1)


class SMPLOutput(SimpleNamespace):
    """Attribute- and key-addressable result (the reference's ModelOutput dataclass, smplx/utils.py:26-62)."""

    def __getitem__(self, k):
        return getattr(self, k)

    def get(self, k, default=None):
        return getattr(self, k, default)

    def keys(self):
        return self.__dict__.keys()


def _as_f32(a):
    if "scipy.sparse" in str(type(a)):
        a = a.todense()
    # C order: np.array keeps the layout of a transposed view (posedirs), and a strided buffer is copied by every kernel call
    # that takes it (17 MB per call for posedirs)
    return torch.from_numpy(np.ascontiguousarray(np.array(a, dtype=np.float32)))


def small_matmul(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A[..., i, k] @ B[..., k, j] for tiny trailing dims as broadcast multiply + sum.  The library's batched GEMM spends
    about 1 ms per call on the 110 k 3x3 / 4x4 products of a 16-frame training step (forward and again in backward)."""
    return (A.unsqueeze(-1) * B.unsqueeze(-3)).sum(-2)


def small_matvec(A: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """A[..., i, k] @ v[..., k] the same way."""
    return (A * v.unsqueeze(-2)).sum(-1)


def rodrigues(rv: torch.Tensor) -> torch.Tensor:
    """[n,3] axis-angle -> [n,3,3]; angle = |rv + 1e-8| as in smplx/lbs.py:316."""
    theta = (rv + 1e-8).norm(dim=1, keepdim=True)
    k = rv / theta
    zero = torch.zeros_like(k[:, 0])
    K = torch.stack([zero, -k[:, 2], k[:, 1], k[:, 2], zero, -k[:, 0], -k[:, 1], k[:, 0], zero], 1).view(-1, 3, 3)
    s, c = torch.sin(theta)[..., None], torch.cos(theta)[..., None]
    return torch.eye(3, dtype=rv.dtype, device=rv.device) + s * K + (1 - c) * (K @ K)


_LEVEL_CACHE = {}


def _tree_plan(par, device):
    """Index tensors of the level-wise walk, built once per (tree, device): [(ids, parent positions or None)], inverse order."""
    key = (tuple(par), str(device))
    if key not in _LEVEL_CACHE:
        order, steps, prev_n = [0], [], 1
        for ids, pos in _tree_levels(par):
            same = len(pos) == prev_n and pos == list(range(len(pos)))
            steps.append((torch.tensor(ids, device=device), None if same else torch.tensor(pos, device=device)))
            order += ids
            prev_n = len(ids)
        inv = [0] * len(par)
        for k, j in enumerate(order):
            inv[j] = k
        _LEVEL_CACHE[key] = (steps, torch.tensor(inv, device=device))
    return _LEVEL_CACHE[key]


def _tree_levels(par):
    """Joints grouped by depth: [(joint ids, position of each joint's parent inside the previous level)]."""
    depth = [0] * len(par)
    for j in range(1, len(par)):
        depth[j] = depth[par[j]] + 1
    levels, prev = [], [0]
    for d in range(1, max(depth) + 1):
        ids = [j for j in range(len(par)) if depth[j] == d]
        levels.append((ids, [prev.index(par[j]) for j in ids]))
        prev = ids
    return levels


def rigid_chain(R: torch.Tensor, J: torch.Tensor, parents: torch.Tensor, parents_host=None):
    """World joint transforms along the kinematic tree, and the same relative to the rest
    pose (smplx/lbs.py:348-404).  R[B,J,3,3], J[B,J,3] -> posed[B,J,3], A[B,J,4,4].
    The tree is walked level by level (9 levels for SMPL's 24 joints) with one batched product per level instead of
    one per joint: a third of the launches, forward and backward."""
    B, nj = J.shape[:2]
    par = parents_host if parents_host is not None else parents.tolist()      # (host copy: no device read per call)
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    local = torch.zeros(B, nj, 4, 4, dtype=R.dtype, device=R.device)
    local[..., :3, :3] = R
    local[..., :3, 3] = rel
    local[..., 3, 3] = 1
    steps, inv = _tree_plan(par, R.device)
    chunks = [local[:, 0:1]]
    for idx, pos in steps:
        up = chunks[-1] if pos is None else chunks[-1].index_select(1, pos)
        chunks.append(small_matmul(up, local.index_select(1, idx)))
    world = torch.cat(chunks, 1).index_select(1, inv)
    rest = torch.cat([J, torch.zeros_like(J[..., :1])], -1)[..., None]      # [B,J,4,1], w = 0
    A = world.clone()
    A[..., :, 3:4] = world[..., :, 3:4] - small_matmul(world, rest)
    return world[..., :3, 3], A


class SMPL(nn.Module):
    NUM_JOINTS = 23
    NUM_BODY_JOINTS = 23

    def __init__(self, model_path: Optional[str] = None, data_struct=None, num_betas: int = 10,
                 batch_size: int = 1, gender: str = "neutral", dtype=torch.float32, **kwargs):
        super().__init__()
        self.gender = gender
        self.batch_size = batch_size
        self.dtype = dtype
        if data_struct is None:
            path = model_path
            if os.path.isdir(model_path):
                path = os.path.join(model_path, f"SMPL_{gender.upper()}.pkl")
            if not os.path.exists(path):
                raise FileNotFoundError(f"Path {path} does not exist!")
            with open(path, "rb") as fh:
                data_struct = pickle.load(fh, encoding="latin1")
        if not isinstance(data_struct, dict):
            data_struct = data_struct.as_pickle_dict() if hasattr(data_struct, "as_pickle_dict") else vars(data_struct)
        d = data_struct
        shapedirs = _as_f32(d["shapedirs"])[:, :, :min(num_betas, 10 if d["shapedirs"].shape[-1] < 300 else 300)]
        self._num_betas = shapedirs.shape[-1]
        self.faces = np.asarray(d["f"])
        self.register_buffer("shapedirs", shapedirs)
        self.vertex_joint_selector = _JointSelector()
        self.register_buffer("faces_tensor", torch.from_numpy(np.array(d["f"], dtype=np.int64)))
        self.betas = nn.Parameter(torch.zeros(batch_size, self._num_betas, dtype=dtype))
        self.global_orient = nn.Parameter(torch.zeros(batch_size, 3, dtype=dtype))
        self.body_pose = nn.Parameter(torch.zeros(batch_size, self.NUM_BODY_JOINTS * 3, dtype=dtype))
        self.transl = nn.Parameter(torch.zeros(batch_size, 3, dtype=dtype))
        self.register_buffer("v_template", _as_f32(d["v_template"]))
        self.register_buffer("J_regressor", _as_f32(d["J_regressor"]))
        pd = np.asarray(d["posedirs"])
        self.register_buffer("posedirs", _as_f32(pd.reshape(-1, pd.shape[-1]).T))
        parents = torch.from_numpy(np.array(d["kintree_table"][0], dtype=np.float32)).long()
        parents[0] = -1
        self.register_buffer("parents", parents)
        self._parents_host = parents.tolist()
        self.register_buffer("lbs_weights", _as_f32(d["weights"]))

    @property
    def num_betas(self):
        return self._num_betas

    def get_num_verts(self):
        return self.v_template.shape[0]

    def forward(self, betas=None, body_pose=None, global_orient=None, transl=None, return_verts=True,
                return_full_pose=False, **kwargs) -> SMPLOutput:
        given = [v for v in (betas, global_orient, body_pose, transl) if v is not None]
        B = max([1] + [len(v) for v in given])
        global_orient = global_orient if global_orient is not None else self.global_orient.expand(B, -1)
        body_pose = body_pose if body_pose is not None else self.body_pose.expand(B, -1)
        betas = betas if betas is not None else self.betas.expand(B, -1)
        if transl is None:
            transl = self.transl
        B = max(betas.shape[0], global_orient.shape[0], body_pose.shape[0])
        if betas.shape[0] != B:
            betas = betas.expand(B, -1)
        pose = torch.cat([global_orient, body_pose], 1)

        needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (betas, pose, transl))
        if self.v_template.is_cuda and not needs_grad:
            # per-frame fast path: three HIP launches (anr_smpl_forward) instead of ~100 framework kernels
            from . import ops
            tr = transl.expand(B, -1) if transl.shape[0] != B else transl
            verts, Jp, A, T, shape_off, pose_off = ops.smpl_forward(
                betas.detach(), pose.detach(), tr.detach(), self.v_template, self.shapedirs, self.posedirs,
                self.J_regressor, self.parents, self.lbs_weights)
            joints = self.vertex_joint_selector(verts, Jp)
            return SMPLOutput(vertices=verts if return_verts else None, joints=joints, betas=betas,
                              global_orient=global_orient, body_pose=body_pose,
                              full_pose=pose if return_full_pose else None,
                              joints_transform=A, vertices_transform=T,
                              shape_offsets=shape_off, pose_offsets=pose_off)

        shape_off = torch.einsum("bl,vcl->bvc", betas, self.shapedirs)
        v_shaped = self.v_template + shape_off
        J = torch.einsum("jv,bvc->bjc", self.J_regressor, v_shaped)
        R = rodrigues(pose.reshape(-1, 3)).view(B, -1, 3, 3)
        feat = (R[:, 1:] - torch.eye(3, dtype=R.dtype, device=R.device)).reshape(B, -1)
        pose_off = (feat @ self.posedirs).view(B, -1, 3)
        v_posed = v_shaped + pose_off
        Jp, A = rigid_chain(R, J, self.parents, self._parents_host)
        nj = self.J_regressor.shape[0]
        # skinning as ONE [V,J] x [J, 16 B] GEMM (a broadcast batched matmul would be B products 16 columns wide)
        nv = self.lbs_weights.shape[0]
        T = (self.lbs_weights @ A.view(B, nj, 16).permute(1, 0, 2).reshape(nj, B * 16)).view(nv, B, 4, 4).permute(1, 0, 2, 3)
        verts = small_matvec(T[..., :3, :3], v_posed) + T[..., :3, 3]
        joints = self.vertex_joint_selector(verts, Jp)

        joints = joints + transl[:, None]
        verts = verts + transl[:, None]
        A = A.clone()
        T = T.clone()
        A[..., :3, 3] += transl[:, None]
        T[..., :3, 3] += transl[:, None]
        return SMPLOutput(vertices=verts if return_verts else None, joints=joints, betas=betas,
                          global_orient=global_orient, body_pose=body_pose,
                          full_pose=pose if return_full_pose else None,
                          joints_transform=A, vertices_transform=T,
                          shape_offsets=shape_off, pose_offsets=pose_off)


def create(model_path: str, model_type: str = "smpl", **kwargs) -> SMPL:
    """smplx/body_models.py:2395-2457, SMPL only (every shipped config has model_type: smpl)."""
    if model_path is not None and os.path.isdir(model_path):
        model_path = os.path.join(model_path, model_type)
    elif model_path is not None:
        model_type = os.path.basename(model_path).split("_")[0].lower()
    if model_type.lower() != "smpl":
        raise ValueError(f"Unknown model type {model_type}, exiting!")
    return SMPL(model_path, **kwargs)
