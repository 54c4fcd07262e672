#!/usr/bin/env python3
"""Disassembly of the gfx950 code objects bundled in a .o / .so:  python3 tools/disasm.py file.o [out.s]"""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_isa import device_code_objects, OBJDUMP  # noqa: E402

path = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
txt = ""
for blob in device_code_objects(path):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(blob)
        f.flush()
        txt += subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
if out:
    open(out, "w").write(txt)
else:
    sys.stdout.write(txt)
