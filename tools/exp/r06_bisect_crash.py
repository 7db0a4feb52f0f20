#!/usr/bin/env python3
"""Which earlier test of tests/test_gpu_training.py makes test_replays_of_one_step_reproduce_its_gradients die in graph.replay()
when the file runs in one process?  Bisects the list of preceding tests (each probe = one pytest process)."""
import subprocess, sys
FILE = "tests/test_gpu_training.py"
TARGET = "test_replays_of_one_step_reproduce_its_gradients"
ids = subprocess.run([sys.executable, "-m", "pytest", FILE, "--collect-only", "-q"], capture_output=True, text=True).stdout.split("\n")
ids = [i for i in ids if "::" in i]
first = min(k for k, i in enumerate(ids) if TARGET in i)
before, target = ids[:first], [i for i in ids if TARGET in i]
def crashes(sub):
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + sub + target, capture_output=True, text=True)
    bad = "Segmentation fault" in r.stdout + r.stderr or r.returncode < 0 or r.returncode > 1
    print(f"  {len(sub):3d} tests + target -> rc {r.returncode} {'CRASH' if bad else 'ok'}", flush=True)
    return bad
print(len(before), "tests precede the target")
if not crashes(before):
    print("the full prefix does not crash here"); sys.exit(0)
lo = before
while len(lo) > 1:
    a, b = lo[:len(lo) // 2], lo[len(lo) // 2:]
    if crashes(a): lo = a
    elif crashes(b): lo = b
    else:
        print("needs tests from both halves:"); break
print("minimal set found:"); [print("  ", t) for t in lo]
