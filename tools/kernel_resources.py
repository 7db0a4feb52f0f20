#!/usr/bin/env python3
"""Compile one HIP source for gfx950 with -save-temps and print per-kernel VGPR/AGPR/spill/LDS
numbers plus mnemonic counts.  Usage: tools/kernel_resources.py anim-nerf_amd/csrc/mlp.hip [substr]"""
import collections, os, re, subprocess, sys, tempfile

src = os.path.abspath(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp(prefix="anr_res_")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps",
                    "-c", src, "-o", os.path.join(tmp, "x.o")], cwd=tmp, capture_output=True, text=True)
if r.returncode:
    print(r.stderr); sys.exit(1)
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
txt = open(os.path.join(tmp, asm)).read()
heads = list(re.finditer(r"^(_Z\w+):", txt, re.M))
for hi, m in enumerate(heads):
    name = m.group(1)
    chunk = txt[m.end():heads[hi + 1].start() if hi + 1 < len(heads) else len(txt)]
    end = re.search(r"\.Lfunc_end\d+:", chunk)
    if not end:
        continue
    body, tail = chunk[:end.start()], chunk[end.end():]
    if filt not in name:
        continue
    ops = collections.Counter(re.findall(r"^\s+([a-z_0-9]+)\s", body, re.M))
    def stat(k):
        mm = re.search(rf"; {k}:? =? ?(\d+)", tail)
        return mm.group(1) if mm else "?"
    print(f"{name}\n   vgpr={stat('NumVgprs')} agpr={stat('NumAgprs')} sgpr={stat('NumSgprs')} scratch={stat('ScratchSize')} "
          f"code={stat('codeLenInByte')}B occupancy={stat('Occupancy')}")
    keys = ["v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x2_f32", "ds_read_b128", "ds_write_b128", "global_load_lds_dwordx4",
            "scratch_load_dword", "scratch_store_dword", "scratch_load_dwordx4", "scratch_store_dwordx4", "scratch_load_dwordx2",
            "scratch_store_dwordx2", "scratch_load_dwordx3", "scratch_store_dwordx3", "v_accvgpr_read_b32", "v_accvgpr_write_b32",
            "v_mov_b32", "s_barrier", "s_waitcnt", "v_cvt_pk_bf16_f32", "s_nop"]
    print("   " + " ".join(f"{k}={ops[k]}" for k in keys if ops[k]))
print("asm:", os.path.join(tmp, asm))
