#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of every kernel in a gfx950 code object (the AMDGPU metadata note).

  python tools/kernel_regs.py [julius_amd/libjulius_amd.so | file.o] [--filter beam_exact] [--json out.json]

Unbundles the gfx950 code object (clang-offload-bundler for .o files, the .hip_fatbin section for the shared library)
and prints, per kernel: VGPRs, AGPRs, SGPRs, SGPR spills, VGPR spills, scratch bytes per thread, static LDS bytes, instruction
count (llvm-objdump)."""
import argparse
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def code_objects(path: Path, td: Path):
    """llvm-objdump --offloading writes every embedded bundle entry next to its input: work on a copy in `td`."""
    import shutil
    cp = td / path.name
    shutil.copy(path, cp)
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(cp)], capture_output=True, text=True, cwd=td)
    return sorted(td.glob("*gfx950*"))


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines()


def kernels_of(co: Path):
    txt = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
    recs = []
    cur = None
    for line in txt.splitlines():
        m = re.match(r"\s*-\s+\.agpr_count:\s+(\d+)", line) or None
        if re.match(r"\s+- \.", line) and cur is not None and ".symbol" in cur:
            recs.append(cur)
            cur = None
        mm = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)$", line)
        if mm:
            if line.lstrip().startswith("- .") and re.match(r"\s{2}- \.", line):
                if cur is not None and ".symbol" in cur:
                    recs.append(cur)
                cur = {}
            if cur is not None:
                cur["." + mm.group(1)] = mm.group(2).strip().strip("'")
    if cur is not None and ".symbol" in cur:
        recs.append(cur)
    # instruction counts
    dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", str(co)], capture_output=True, text=True).stdout
    counts, name = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            counts[name] = {"insts": 0, "readlane": 0, "writelane": 0, "scratch": 0, "flat": 0}
            continue
        if name and re.match(r"^\s+[a-z_0-9]+", line):
            c = counts[name]
            c["insts"] += 1
            op = line.split()[0]
            if op.startswith("v_readlane"):
                c["readlane"] += 1
            elif op.startswith("v_writelane"):
                c["writelane"] += 1
            elif op.startswith("scratch_"):
                c["scratch"] += 1
            elif op.startswith("flat_"):
                c["flat"] += 1
    out = []
    for r in recs:
        if ".sgpr_count" not in r:
            continue
        sym = r[".symbol"].replace(".kd", "")
        out.append({"symbol": sym, "vgpr": int(r.get(".vgpr_count", 0)), "agpr": int(r.get(".agpr_count", 0)),
                    "sgpr": int(r.get(".sgpr_count", 0)), "sgpr_spill": int(r.get(".sgpr_spill_count", 0)),
                    "vgpr_spill": int(r.get(".vgpr_spill_count", 0)), "scratch_bytes": int(r.get(".private_segment_fixed_size", 0)),
                    "lds_static": int(r.get(".group_segment_fixed_size", 0)), **counts.get(sym, {})})
    names = demangle([k["symbol"] for k in out])
    for k, n in zip(out, names):
        n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
        # template arguments kept, the parameter list dropped
        depth, cut = 0, len(n)
        for i, ch in enumerate(n):
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = i
                break
        k["name"] = n[:cut]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path", nargs="?", default=str(Path(__file__).resolve().parent.parent / "julius_amd" / "libjulius_amd.so"))
    ap.add_argument("--filter", default=None)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        ks = []
        for co in code_objects(Path(a.path), Path(td)):
            ks += kernels_of(co)
    if a.filter:
        ks = [k for k in ks if a.filter in k["name"]]
    ks.sort(key=lambda k: k["name"])
    print(f"{'kernel':70s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'s_spill':>7s} {'v_spill':>7s} {'scratch':>7s} {'lds':>6s} {'insts':>6s} {'rd/wrlane':>9s}")
    for k in ks:
        print(f"{k['name'][:70]:70s} {k['vgpr']:4d} {k['agpr']:4d} {k['sgpr']:4d} {k['sgpr_spill']:7d} {k['vgpr_spill']:7d} {k['scratch_bytes']:7d} "
              f"{k['lds_static']:6d} {k.get('insts', 0):6d} {k.get('readlane', 0) + k.get('writelane', 0):9d}")
    if a.json:
        Path(a.json).write_text(json.dumps(ks, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
