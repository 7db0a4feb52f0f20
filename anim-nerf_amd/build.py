"""Builds libanimnerf_hip.so (gfx950) in-tree with hipcc.  No CPU fallback is ever built."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libanimnerf_hip.so")
SOURCES = ["frame_ops.hip", "frame_setup.hip", "frame_bwd.hip", "smpl.hip", "composite.hip", "compact.hip", "warp.hip", "knn_k.hip", "mlp.hip", "mlp_bwd.hip", "mlp_wgrad.hip", "train_glue.hip", "train_step.hip", "mesh.hip", "mlp_inst_f32.hip", "mlp_inst_f32_train.hip",
           "mlp_inst_bf16.hip", "mlp_inst_bf16_train.hip", "mlp_inst_pre.hip", "mlp_inst_tan.hip", "mlp_inst_view.hip", "mlp_inst_refine.hip"]
HEADERS = ["anr_common.h", "mlp_core.h", os.path.join("..", "..", "include", "animnerf_hip.h")]


# The 4-wave x 64-point bf16 variant needs more than 256 registers per lane: with the accumulators in AGPRs (hipcc's
# default) its epilogue spends 2,700 v_accvgpr_read per point tile; VGPR-form MFMAs cut that to 500 (51 -> 54 % of peak).
PER_SOURCE_FLAGS = {"mlp_inst_bf16_train.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
# No SLP vectorisation: the packed fp32 forms it produces (v_pk_add_f32 / v_pk_mul_f32 with op_sel and neg modifiers, one half
# of the pair swapped) sporadically returned the unmodified operand in one lane when other kernels shared the GPU — the step's
# parallel branches — 1-2 % of replayed steps with a wrong red channel or a pose gradient off by 1e-3..1e-2 (DESIGN 4.4,
# tools/exp/race_hunt.py, profiles/r05/race_hunt_*.txt).  Without it: 0 deviations above 1.2e-6 in 8,000 replays, and
# no kernel is slower (profiles/r05/ab_no_slp.txt).  ANR_BUILD_SLP=1 builds the old way, for the hunting tool.
NO_SLP = [] if os.environ.get("ANR_BUILD_SLP") else ["-fno-slp-vectorize"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, defines=(), out: str = None, extra_flags=()) -> str:
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path.
    `defines`/`out` build an experiment variant (e.g. timing ablations) next to the product library."""
    lib_path = out or LIB_PATH
    if not defines and not out and not force and not extra_flags and not _stale():
        return LIB_PATH
    objs = []
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
             "-Wno-unused-value", "-Wno-pass-failed"] + [f"-D{d}" for d in defines] + list(extra_flags)
    flags += NO_SLP
    tag = ("." + "_".join(defines)) if defines else ""
    if extra_flags:
        tag += ".x" + hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8]
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", tag + ".o"))
        cmd = [_hipcc(), *flags, *PER_SOURCE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return lib_path


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[6:] for a in sys.argv[1:] if a.startswith("--out=")]
    extra = [f for a in sys.argv[1:] if a.startswith("--flag=") for f in a[7:].split(",")]
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, defines=defs, out=outs[0] if outs else None,
                extra_flags=extra))
