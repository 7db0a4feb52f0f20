#!/usr/bin/env python3
"""tests/golden/mocap.npz: the reference's `load_mixamo_smpl` (/root/reference/novel_pose.py:26-41) RUN on a small synthetic
`result.pkl`.  novel_pose.py itself imports pyrender / trimesh / imageio / Lightning (none installed), so the one function is
taken out of the reference file's syntax tree and executed with the reference's own `load_pickle_file` — build container only;
the fixture holds the inputs and the function's outputs.      python tests/golden/make_mocap_fixture.py"""
import ast
import os
import pickle
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mocap.npz")
import importlib.util                                          # noqa: E402
_spec = importlib.util.spec_from_file_location("ref_util", os.path.join(REF, "utils", "util.py"))   # (utils/__init__.py imports cv2)
_util = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_util)
load_pickle_file = _util.load_pickle_file

tree = ast.parse(open(os.path.join(REF, "novel_pose.py")).read())
fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "load_mixamo_smpl")
ns = {"os": os, "np": np, "load_pickle_file": load_pickle_file}
exec(compile(ast.Module(body=[fn], type_ignores=[]), "novel_pose.py", "exec"), ns)

rng = np.random.RandomState(5)
anim_len, skip = 7, 2
smpl_array = rng.randn(anim_len * 72).astype(np.float32)
cam_array = rng.randn(anim_len, 3).astype(np.float32)
tmp = tempfile.mkdtemp()
os.makedirs(os.path.join(tmp, "0007"))
with open(os.path.join(tmp, "0007", "result.pkl"), "wb") as f:
    pickle.dump({"anim_len": anim_len, "smpl_array": smpl_array, "cam_array": cam_array}, f)
mocap = ns["load_mixamo_smpl"](tmp, "0007", skip)
np.savez_compressed(OUT, anim_len=anim_len, skip=skip, smpl_array=smpl_array, cam_array=cam_array,
                    global_orient=np.stack([m["global_orient"] for m in mocap]), body_pose=np.stack([m["body_pose"] for m in mocap]),
                    transl=np.stack([m["transl"] for m in mocap]), cam=np.stack([m["cam"] for m in mocap]))
print("mocap.npz:", len(mocap), "frames")
