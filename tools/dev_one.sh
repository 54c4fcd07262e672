#!/bin/bash
# Register allocation of ONE kernel instantiation, compiled alone (seconds):
#   tools/dev_one.sh 'k_compress<double, 1, true, 2, 0>' 'FwdParams<double>' [extra hipcc flags]
K=$1; A=$2; shift 2
cd "$(dirname "$0")/../dctz_amd"
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -Wno-pass-failed -I../include --cuda-device-only -c csrc/dctz_kernels.hip -o /tmp/dev_one.o "-DDCTZ_DEV_ONE=$K" "-DDCTZ_DEV_ARGS=$A" || exit 1
python3 - <<PY
import subprocess, re
import sys; sys.path.insert(0, "../tools"); from check_isa import device_code_objects
blobs = device_code_objects("/tmp/dev_one.o") or [open("/tmp/dev_one.o", "rb").read()]
open("/tmp/dev_one.co", "wb").write(blobs[0])
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", "/tmp/dev_one.co"], capture_output=True, text=True).stdout
for k in ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
    m = re.search(r"\.%s:\s*(\d+)" % k, txt)
    print(k, m.group(1) if m else "?")
PY
