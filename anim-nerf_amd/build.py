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
           "mlp_inst_bf16.hip", "mlp_inst_bf16_train.hip", "mlp_inst_pre.hip", "mlp_inst_tan.hip", "mlp_inst_view.hip", "mlp_inst_refine.hip",
           "ray_march.hip"]
HEADERS = ["anr_common.h", "mlp_core.h", "composite_core.h", "warp_core.h", os.path.join("..", "..", "include", "animnerf_hip.h")]


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


_LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def packed_fp32_forms(lib_path: str = LIB_PATH):
    """Disassemble every gfx950 code object inside `lib_path` and list the packed fp32 VALU instructions
    (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) that carry operand modifiers, as (kernel, instruction text, suspicious).
    suspicious = the form the library must not contain: an `op_sel` that swaps the halves of an operand, or a
    negation of ONE half only — what the SLP vectoriser makes of `r += 1 - w; g += 1 - w` and what returned the unmodified
    operand in one lane next to other kernels.  (The explicit float2 arithmetic of the neighbour search compiles to plain
    both-halves subtractions, neg_lo == neg_hi with at most an op_sel_hi broadcast; every warp test compares its results bit
    for bit.)  The flag -fno-slp-vectorize is the means; this scan of the generated ISA is the gate (build() raises)."""
    import re
    import tempfile
    objcopy, bundler, objdump = (os.path.join(_LLVM_BIN, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump"))
    found = []
    with tempfile.TemporaryDirectory(prefix="anr_isa_") as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([objcopy, f"--dump-section=.hip_fatbin={fat}", lib_path, os.path.join(tmp, "copy.so")], check=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
        if not starts:
            raise RuntimeError(f"{lib_path}: no offload bundle in .hip_fatbin")
        pk = re.compile(r"\bv_pk_(?:add|mul|fma)_f32\b")
        mod = re.compile(r"(op_sel_hi|op_sel|neg_lo|neg_hi):\[([\d,]+)\]")
        for i, o in enumerate(starts):
            piece, co = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"co{i}.o")
            open(piece, "wb").write(data[o:starts[i + 1] if i + 1 < len(starts) else len(data)])
            subprocess.run([bundler, "--unbundle", "--type=o", f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--output={co}"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            text = subprocess.run([objdump, "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout
            kernel = "?"
            for line in text.splitlines():
                head = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if head:
                    kernel = head.group(1)
                    continue
                if not pk.search(line):
                    continue
                mods = dict(mod.findall(line))
                if not mods:
                    continue
                swapped = "1" in mods.get("op_sel", "")
                lopsided = mods.get("neg_lo", "") != mods.get("neg_hi", "")
                found.append((kernel, line.split("//")[0].strip(), swapped or lopsided))
    return found


def check_isa(lib_path: str = LIB_PATH) -> int:
    """Raise if the linked library holds a packed fp32 instruction of the suspicious form; returns the number of (benign)
    modified packed instructions it does hold."""
    forms = packed_fp32_forms(lib_path)
    bad = [f for f in forms if f[2]]
    if bad:
        lines = "\n".join(f"  {k}: {t}" for k, t, _ in bad[:20])
        raise RuntimeError(f"{lib_path}: {len(bad)} packed fp32 instruction(s) with swapped / half-negated operands "
                           f"(DESIGN 4.4: build with -fno-slp-vectorize, write such pairs as scalars):\n{lines}")
    return len(forms)


def _deps(src: str):
    """`src` and the project headers it includes, transitively (paths)."""
    import re
    seen, todo = [], [os.path.join(CSRC, src)]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f).read(), re.M):
            for base in (os.path.dirname(f), CSRC, os.path.join(HERE, "..", "include")):
                if os.path.exists(os.path.join(base, inc)):
                    todo.append(os.path.join(base, inc))
                    break
    return seen


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
        objs.append(obj)
        cmd = [_hipcc(), *flags, *PER_SOURCE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        # an object is kept when it is newer than its source, the headers that source includes (by name, one level of nesting
        # is all there is) and this recipe, and was made by the same command line (recorded next to it)
        stamp = obj + ".cmd"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == " ".join(cmd):
            deps = _deps(src) + [os.path.abspath(__file__)]
            if all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
                continue
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), stamp, " ".join(cmd)))
    failed = []
    for src, p, stamp, cmdline in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"hipcc failed on {src}:\n{out}")
            continue
        open(stamp, "w").write(cmdline)
        if verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("\n".join(failed))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if not os.environ.get("ANR_BUILD_SLP"):                     # (the hunting tool builds the old way on purpose)
        try:
            check_isa(lib_path)
        except RuntimeError:
            os.replace(lib_path, lib_path + ".rejected")       # a library that fails the gate is not left where it would load
            raise
    return lib_path


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a[6:] for a in sys.argv[1:] if a.startswith("--out=")]
    extra = [f for a in sys.argv[1:] if a.startswith("--flag=") for f in a[7:].split(",")]
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, defines=defs, out=outs[0] if outs else None,
                extra_flags=extra))
