#!/usr/bin/env python3
"""Per-kernel resource table of the built gfx950 code objects (VGPRs, AGPRs, SGPRs, LDS, scratch, spills), read from the
code object's metadata notes -- the check that a refactor of the big kernels has not changed their register allocation
(occupancy is decided there: k_compress<double> must stay at two waves per SIMD without scratch).

  python3 tools/kernel_resources.py [lib.so|obj.o ...] [--filter k_compress] [--json]
"""
import json
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_isa import device_code_objects  # noqa: E402

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
FILT = "c++filt"


def kernels_of(path):
    out = []
    for blob in device_code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "agpr_count":
                cur = {"agpr": int(v)}
                out.append(cur)
            elif cur is not None:
                if k == "name":
                    cur["name"] = v
                elif k in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                           "vgpr_spill_count", "sgpr_spill_count"):
                    cur[k] = int(v)
    names = [k.get("name", "?") for k in out]
    if names:
        dem = subprocess.run([FILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        for k, d in zip(out, dem):
            k["demangled"] = re.sub(r"^void ", "", d)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    filt = None
    if "--filter" in sys.argv:
        filt = sys.argv[sys.argv.index("--filter") + 1]
        args = [a for a in args if a != filt]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = args or [os.path.join(root, "dctz_amd", "lib", "libdctzhip.so")]
    rows = []
    for p in paths:
        for k in kernels_of(p):
            if filt and filt not in k.get("demangled", ""):
                continue
            rows.append(k)
    rows.sort(key=lambda k: k.get("demangled", ""))
    if "--json" in sys.argv:
        print(json.dumps(rows, indent=1))
        return
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>8} {'vspill':>7} {'sspill':>7}  kernel")
    for k in rows:
        nm = re.sub(r"dctz::", "", k.get("demangled", k.get("name", "?")))
        nm = re.sub(r"\(.*$", "", nm)
        print(f"{k.get('vgpr_count', 0):>5} {k.get('agpr', 0):>5} {k.get('sgpr_count', 0):>5} {k.get('group_segment_fixed_size', 0):>7} "
              f"{k.get('private_segment_fixed_size', 0):>8} {k.get('vgpr_spill_count', 0):>7} {k.get('sgpr_spill_count', 0):>7}  {nm}")


if __name__ == "__main__":
    main()
