"""Seeded synthetic world for the Anim-NeRF rendering path.

The licensed SMPL model files, the People-Snapshot frames and the released
checkpoints are not available offline (reference README.md:45-56, :110), so the
tests, the oracle fixtures and bench.py all run on a synthetic SMPL-like body
table with the real model's shapes (V=6890 vertices, J=24 joints, 10 shape
and 207 pose blend-shape directions) and the real SMPL kinematic tree.

Everything here is numpy + a PCG64 generator: the same seed gives the same
tables on every machine.  `table_checksum` lets a fixture refuse to compare
against tables that were regenerated differently.
"""
from __future__ import annotations

import hashlib
import pickle
from dataclasses import dataclass

import numpy as np

NUM_VERTS = 6890
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_BASIS = 207
NUM_FACES = 13776

# SMPL kinematic tree (parent of each joint; root = -1).
SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
    dtype=np.int64)

# Rest-pose joint locations of the synthetic body (metres, y up, facing +z).
_REST_JOINTS = np.array([
    [0.000, -0.220, 0.000], [0.070, -0.310, 0.000], [-0.070, -0.310, 0.000],
    [0.000, -0.110, -0.020], [0.100, -0.690, 0.000], [-0.100, -0.690, 0.000],
    [0.000, 0.030, 0.000], [0.090, -1.090, -0.030], [-0.090, -1.090, -0.030],
    [0.000, 0.080, 0.000], [0.110, -1.150, 0.090], [-0.110, -1.150, 0.090],
    [0.000, 0.290, -0.030], [0.080, 0.200, -0.020], [-0.080, 0.200, -0.020],
    [0.000, 0.380, 0.020], [0.180, 0.230, -0.030], [-0.180, 0.230, -0.030],
    [0.440, 0.220, -0.040], [-0.440, 0.220, -0.040], [0.690, 0.220, -0.040],
    [-0.690, 0.220, -0.040], [0.780, 0.210, -0.050], [-0.780, 0.210, -0.050],
], dtype=np.float64)

# Capsule radius around the bone that ENDS at joint c (parent(c) -> c).
_BONE_RADIUS = np.array([
    0.00, 0.085, 0.085, 0.120, 0.070, 0.070, 0.125, 0.050, 0.050, 0.130, 0.040, 0.040,
    0.060, 0.070, 0.070, 0.085, 0.055, 0.055, 0.045, 0.045, 0.035, 0.035, 0.030, 0.030,
], dtype=np.float64)


def _smoothstep(x):
    x = np.clip(x, 0.0, 1.0)
    return x * x * (3.0 - 2.0 * x)


def _orthobasis(axis):
    """Two unit vectors orthogonal to `axis` (exact ops only)."""
    a = axis / np.sqrt(np.sum(axis * axis))
    helper = np.array([0.0, 0.0, 1.0]) if abs(a[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
    u = np.cross(a, helper)
    u = u / np.sqrt(np.sum(u * u))
    v = np.cross(a, u)
    return a, u, v


@dataclass
class SyntheticSMPL:
    """Arrays in the layout of an SMPL pickle (reference smplx/body_models.py:125-251)."""
    v_template: np.ndarray      # [V,3] f32
    shapedirs: np.ndarray       # [V,3,10] f32
    posedirs: np.ndarray        # [V,3,207] f32
    J_regressor: np.ndarray     # [24,V] f32
    weights: np.ndarray         # [V,24] f32
    kintree_table: np.ndarray   # [2,24] i64
    f: np.ndarray               # [F,3] i64

    def as_pickle_dict(self):
        return {
            "v_template": self.v_template, "shapedirs": self.shapedirs,
            "posedirs": self.posedirs, "J_regressor": self.J_regressor,
            "weights": self.weights, "kintree_table": self.kintree_table, "f": self.f,
        }

    def write_pickle(self, path):
        with open(path, "wb") as fh:
            pickle.dump(self.as_pickle_dict(), fh, protocol=2)


def make_smpl_table(seed: int = 0) -> SyntheticSMPL:
    """Vertices on bone capsules, <=4-sparse smooth skinning weights, small
    smooth blend-shape bases.  Vertices of neighbouring bones come within a few
    centimetres of each other (thighs, arm/torso) so the blend-weight
    confidence test of the warp (reference models/anim_nerf.py:165-168) fires
    both ways."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parents = SMPL_PARENTS
    J = _REST_JOINTS

    # --- vertex budget per bone, proportional to capsule area ---------------
    bones = list(range(1, NUM_JOINTS))
    length = np.array([np.sqrt(np.sum((J[c] - J[parents[c]]) ** 2)) for c in bones])
    area = (length + 2.0 * _BONE_RADIUS[bones]) * _BONE_RADIUS[bones]
    counts = np.floor(area / area.sum() * NUM_VERTS).astype(np.int64)
    counts[np.argmax(counts)] += NUM_VERTS - counts.sum()

    verts = np.zeros((NUM_VERTS, 3))
    weights = np.zeros((NUM_VERTS, NUM_JOINTS))
    owner = np.zeros(NUM_VERTS, dtype=np.int64)     # bone (child joint id) of each vertex
    frac = np.zeros(NUM_VERTS)
    n0 = 0
    for bi, c in enumerate(bones):
        p = parents[c]
        n = int(counts[bi])
        a, u, v = _orthobasis(J[c] - J[p])
        # stratified along the bone (overshoot both ends a little = rounded caps)
        t = (np.arange(n) + rng.random(n)) / n * 1.3 - 0.15
        # golden-angle spiral around the bone, via a rational circle parametrisation
        k = np.arange(n) * 0.6180339887498949
        k = k - np.floor(k)
        s = np.where(k < 0.5, 4.0 * k - 1.0, 3.0 - 4.0 * k)           # triangle wave in [-1,1]
        q = np.where(k < 0.5, 1.0, -1.0)
        cth = s
        sth = q * np.sqrt(np.maximum(0.0, 1.0 - s * s))
        r = _BONE_RADIUS[c] * (0.9 + 0.2 * rng.random(n))
        # taper the caps
        cap = np.where(t < 0.0, -t / 0.15, np.where(t > 1.0, (t - 1.0) / 0.15, 0.0))
        r = r * np.sqrt(np.maximum(0.05, 1.0 - cap * cap))
        pos = J[p][None] + t[:, None] * (J[c] - J[p])[None] * 1.0 \
            + (r * cth)[:, None] * u[None] + (r * sth)[:, None] * v[None]
        verts[n0:n0 + n] = pos
        owner[n0:n0 + n] = c
        frac[n0:n0 + n] = t
        # skinning: the segment parent(c)->c moves with joint parent(c)
        w_main = np.ones(n)
        w_up = 0.5 * (1.0 - _smoothstep((t + 0.15) / 0.45))     # towards grandparent near the proximal end
        w_dn = 0.5 * _smoothstep((t - 0.70) / 0.45)             # towards the child joint near the distal end
        gp = parents[p] if parents[p] >= 0 else p
        weights[n0:n0 + n, p] += w_main - w_up - w_dn
        weights[n0:n0 + n, gp] += w_up
        weights[n0:n0 + n, c] += w_dn
        # a sparse fourth influence on ~1/3 of the vertices
        sib = int(rng.integers(0, NUM_JOINTS))
        extra = 0.04 * rng.random(n) * (rng.random(n) < 0.33)
        weights[n0:n0 + n, sib] += extra
        n0 += n
    assert n0 == NUM_VERTS
    weights = np.maximum(weights, 0.0)
    weights = weights / weights.sum(axis=1, keepdims=True)

    # shuffle so that vertex index carries no spatial order (as in a real mesh)
    perm = rng.permutation(NUM_VERTS)
    verts, weights, owner, frac = verts[perm], weights[perm], owner[perm], frac[perm]

    # --- joint regressor: mean of the 40 vertices nearest to each design joint
    J_reg = np.zeros((NUM_JOINTS, NUM_VERTS))
    for j in range(NUM_JOINTS):
        d2 = np.sum((verts - J[j][None]) ** 2, axis=1)
        near = np.argsort(d2, kind="stable")[:40]
        J_reg[j, near] = 1.0 / 40.0

    # --- shape blend shapes: smooth low-order fields + a little noise --------
    x, y, z = verts[:, 0], verts[:, 1], verts[:, 2]
    fields = [
        np.stack([0.04 * x, 0 * y, 0.04 * z], 1),            # girth
        np.stack([0 * x, 0.05 * (y + 0.2), 0 * z], 1),       # height
        np.stack([0.03 * x * (y > 0), 0 * y, 0 * z], 1),     # shoulders
        np.stack([0.03 * x * (y < -0.2), 0 * y, 0.02 * z * (y < -0.2)], 1),  # legs
        np.stack([0 * x, 0 * y, 0.05 * z * (np.abs(x) < 0.2)], 1),           # belly
        np.stack([0.02 * x * y, 0.02 * y * y, 0 * z], 1),
        np.stack([0.02 * z, 0 * y, -0.02 * x], 1),
        np.stack([0 * x, 0.02 * x * x, 0.02 * y * z], 1),
        np.stack([0.01 * y, -0.01 * x, 0.01 * z], 1),
        np.stack([0.015 * x * x, 0.015 * y * z, 0.015 * z * z], 1),
    ]
    shapedirs = np.stack(fields, axis=-1) + 0.001 * (rng.random((NUM_VERTS, 3, NUM_BETAS)) - 0.5)

    # --- pose blend shapes: small, driven mostly by the owning joints ---------
    posedirs = 0.004 * (rng.random((NUM_VERTS, 3, NUM_POSE_BASIS)) - 0.5)
    # boost the 9 basis vectors of the joint that owns each vertex
    for j in range(1, NUM_JOINTS):
        sel = np.where(weights[:, j] > 0.3)[0]
        posedirs[sel, :, (j - 1) * 9:(j - 1) * 9 + 9] *= 4.0

    faces = rng.integers(0, NUM_VERTS, size=(NUM_FACES, 3)).astype(np.int64)
    kintree = np.stack([np.where(parents < 0, 2 ** 32 - 1, parents), np.arange(NUM_JOINTS)]).astype(np.int64)

    return SyntheticSMPL(
        v_template=verts.astype(np.float32),
        shapedirs=shapedirs.astype(np.float32),
        posedirs=posedirs.astype(np.float32),
        J_regressor=J_reg.astype(np.float32),
        weights=weights.astype(np.float32),
        kintree_table=kintree,
        f=faces)


def table_checksum(tbl: SyntheticSMPL) -> str:
    h = hashlib.sha256()
    for a in (tbl.v_template, tbl.shapedirs, tbl.posedirs, tbl.J_regressor, tbl.weights):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------
# poses / cameras
# ---------------------------------------------------------------------------

def template_pose_params() -> dict:
    """Values of the reference's assets/X_pose.pkl (the canonical template pose:
    zeros except the hips, +-0.5 rad about z; read in tools/prepare_template.py:27)."""
    body_pose = np.zeros(69, dtype=np.float32)
    body_pose[2] = 0.5
    body_pose[5] = -0.5
    return {
        "betas": np.zeros((1, 10), np.float32),
        "global_orient": np.zeros((1, 3), np.float32),
        "body_pose": body_pose[None],
        "transl": np.zeros((1, 3), np.float32),
    }


def animated_pose_params(seed: int = 1, bs: int = 1, pose_std: float = 0.2,
                         transl_z: float = -3.0) -> dict:
    """A seeded 'animated' SMPL pose: N(0, pose_std) joint angles, a small global
    rotation, the body pushed to z = transl_z in front of an identity camera."""
    rng = np.random.Generator(np.random.PCG64(seed))

    def gauss(shape):                      # sum of 12 uniforms - 6: exact ops, ~N(0,1)
        return rng.random(shape + (12,)).sum(-1) - 6.0
    body_pose = pose_std * gauss((bs, 69))
    body_pose[:, 2] += 0.5
    body_pose[:, 5] -= 0.5
    transl = np.zeros((bs, 3))
    transl[:, 2] = transl_z
    transl[:, :2] = 0.02 * gauss((bs, 2))
    return {
        "betas": (0.5 * gauss((1, 10))).repeat(bs, 0).astype(np.float32),
        "global_orient": (0.15 * gauss((bs, 3))).astype(np.float32),
        "body_pose": body_pose.astype(np.float32),
        "transl": transl.astype(np.float32),
    }


def static_pose_params(bs: int = 1, transl_z: float = -3.0) -> dict:
    """Pose == template pose (the 'fixed SMPL, no warp' configuration), body at z=transl_z."""
    p = template_pose_params()
    out = {k: np.repeat(v, bs, 0) for k, v in p.items()}
    out["transl"] = out["transl"].copy()
    out["transl"][:, 2] = transl_z
    return out


def pinhole_camera(H: int, W: int, focal_scale: float = 1.1):
    """Identity-rotation camera at the origin looking down -z."""
    c2w = np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32)
    focal = np.array([focal_scale * W, focal_scale * W], np.float32)
    center = np.array([W * 0.5, H * 0.5], np.float32)
    return c2w, focal, center
