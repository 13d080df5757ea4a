#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures from hipcc's device assembly (the .amdhsa metadata block).

    tools/kernel_resources.py [--build DIR] [--diff OLD_DIR] [file.hip ...]

--build DIR : compile every rustradio_amd/csrc/kernels_*.hip (or the files named) to DIR/<name>.s with the product's flags
              (device only, no GPU needed) and print one line per kernel: VGPRs, AGPRs, SGPRs, spilled VGPRs, scratch bytes
--diff OLD  : print only the kernels whose figures differ from OLD/<name>.s (a build of another revision)

A tile kernel that starts to spill is 1.5-2x slower (every scratch access waits on the whole in-order vmcnt queue,
profiles/TUNING_LOG.md §4.1), so every change to a kernel is checked with this before it goes to the GPU.
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rustradio_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "--cuda-device-only", "-S"]
NO_SLP = {"kernels_fft.hip", "kernels_poly.hip"}


def build(files, out, extra):
    os.makedirs(out, exist_ok=True)
    procs = []
    for f in files:
        name = os.path.basename(f)
        cmd = ["hipcc"] + FLAGS + (["-fno-slp-vectorize"] if name in NO_SLP else []) + extra + \
              ["-o", os.path.join(out, name.replace(".hip", ".s")), f]
        procs.append((name, subprocess.Popen(cmd, cwd=CSRC, stderr=subprocess.PIPE)))
    for name, p in procs:
        _, err = p.communicate()
        if p.returncode:
            sys.exit(f"{name}: {err.decode()[-2000:]}")


def parse(path):
    """-> {demangled-ish kernel name: (vgpr, agpr, sgpr, vgpr_spill, scratch, lds)}"""
    txt = open(path, errors="replace").read()
    out = {}
    for blk in txt.split("  - .agpr_count:")[1:]:
        def g(key):
            m = re.search(r"\." + key + r":\s+(\S+)", blk)
            return m.group(1) if m else "0"
        name = g("name")
        agpr = blk.split("\n", 1)[0].strip()
        out[name] = tuple(int(x) for x in (g("vgpr_count"), agpr, g("sgpr_count"), g("vgpr_spill_count"),
                                             g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    return out


def demangle(names):
    try:
        p = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True)
        return dict(zip(names, p.stdout.split("\n")))
    except Exception:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", default="/tmp/rr_kres")
    ap.add_argument("--diff", default=None)
    ap.add_argument("--extra", default="", help="extra compiler flags, e.g. -DRR_POLY_NB=2")
    ap.add_argument("files", nargs="*")
    a = ap.parse_args()
    files = [os.path.abspath(f) for f in a.files] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith("kernels_") and f.endswith(".hip"))
    build(files, a.build, a.extra.split())
    bad = 0
    for f in files:
        sname = os.path.basename(f).replace(".hip", ".s")
        new = parse(os.path.join(a.build, sname))
        old = parse(os.path.join(a.diff, sname)) if a.diff and os.path.exists(os.path.join(a.diff, sname)) else None
        dm = demangle(list(new))
        for k, v in sorted(new.items(), key=lambda kv: dm[kv[0]]):
            if old is not None and old.get(k) == v:
                continue
            label = re.sub(r"\(.*", "", dm[k]).replace("void rr::", "")
            was = f"   was {old[k]}" if old is not None and k in old else ("   (new)" if old is not None else "")
            flag = "  <-- SPILLS" if v[3] or v[4] else ""
            bad += bool(v[3] or v[4])
            print(f"{label:70s} vgpr {v[0]:3d} agpr {v[1]:3d} sgpr {v[2]:3d} spill {v[3]:3d} scratch {v[4]:4d} B{was}{flag}")
    print(f"{bad} kernels with spills / scratch" if bad else "no kernel spills")


if __name__ == "__main__":
    main()
