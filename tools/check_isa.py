#!/usr/bin/env python3
"""ISA lint of the built gfx950 code object (dctz_amd/lib/libdctzhip.so, or any .so / .o given).

Rules (the round-1 data corruption, reproduced in tools/ubench/probe_r2.hip on MI355X): (A) NO vector-memory store of more
than 64 bits of data -- buffer_ / global_ / flat_ / scratch_, any soffset -- may be followed directly by a VALU
instruction that writes one of its data registers; (B) a MUBUF store of that width whose soffset operand is an SGPR is
refused as a form (and must not be followed by such a write within two instructions).  LLVM's hazard recognizer only inserts the
wait states for the form WITHOUT a register soffset (GCNHazardRecognizer::createsVALUHazard), although the hardware
needs them for both: with no wait state in between, ~0.7 % of such stores carried the NEW register contents.
The kernels therefore never use a register soffset on 16-byte buffer stores; this script checks the result.

Exit code 0: clean; 1: violations (printed); 2: cannot inspect."""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_code_objects(path):
    """gfx950 code objects bundled into a host .so / .o by hipcc."""
    data = open(path, "rb").read()
    out = []
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + len(MAGIC)
    return out


def regs(tok):
    """v12 -> {12}; v[12:15] -> {12..15}; anything else -> empty set."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


WIDE = r"(dwordx3|dwordx4|format_xyzw?|format_d16_xyzw)"


def _valu_writes(op, args):
    """Registers a VALU instruction writes (empty for compares into SGPRs, lane reads, non-VALU)."""
    if not op.startswith("v_") or op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return set()
    return regs(args.split(",")[0].strip())


def lint(disasm):
    """Returns (number of wide VMEM stores seen, list of violations).

    Rule A, every store of more than 64 bits -- buffer_ / global_ / flat_ / scratch_, any soffset: the instruction right
    behind it must not be a VALU write of one of its data registers (tools/ubench/probe_r2.hip: with soffset = 0 and NO
    wait state 19 % of such stores carried the new register contents; with one wait state none).  For the forms LLVM's
    hazard recognizer knows it inserts that wait state itself; this rule verifies the result instead of trusting it.
    Rule B, buffer stores with a REGISTER soffset: the recognizer does not cover them at all, so the form itself is
    refused, and a VALU write of a data register within two instructions is reported on top."""
    ins = []
    func = "?"
    for line in disasm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            func = m.group(1)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*(//.*)?$", line)
        if m and not line.lstrip().startswith("//"):
            ins.append((func, m.group(1), m.group(2)))
    bad, seen = [], 0
    for k, (f, op, args) in enumerate(ins):
        m = re.fullmatch(r"(buffer|global|flat|scratch)_store_" + WIDE, op)
        if not m:
            continue
        seen += 1
        toks = [t.strip() for t in args.split(",")]
        # operands: buffer: vdata, vaddr|off, srsrc, soffset; global / flat / scratch: vaddr|off, vdata, ...
        data = regs(toks[0].split()[0]) if m.group(1) == "buffer" else (regs(toks[1].split()[0]) if len(toks) > 1 else set())
        # Rule A
        if k + 1 < len(ins) and ins[k + 1][0] == f:
            op2, args2 = ins[k + 1][1], ins[k + 1][2]
            if _valu_writes(op2, args2) & data:
                bad.append(f"{f}: '{op} {args}' followed with no wait state by '{op2} {args2}' (a VALU write of its data registers)")
        if m.group(1) != "buffer":
            continue
        soff = toks[3].split()[0] if len(toks) > 3 else ""
        if not re.fullmatch(r"s\d+|m0|vcc_lo|vcc_hi|ttmp\d+", soff):
            continue                                        # immediate / inline constant: covered by rule A
        # Rule B
        for d in (1, 2):
            if k + d >= len(ins) or ins[k + d][0] != f:
                break
            op2, args2 = ins[k + d][1], ins[k + d][2]
            if op2.startswith("s_nop"):
                n = int(args2.split()[0], 0) + 1 if args2 else 1
                if n >= 2 or d == 2:
                    break
                continue
            if d == 2 and _valu_writes(op2, args2) & data:  # (d == 1 is rule A's)
                bad.append(f"{f}: '{op} {args}' followed after 1 instruction(s) by '{op2} {args2}'")
        bad.append(f"{f}: '{op} {args}' uses a register soffset (use soffset = 0 and put the offset into the VGPR / immediate)")
    return seen, bad


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sys.argv[1:] or [os.path.join(root, "dctz_amd", "lib", "libdctzhip.so")]
    total, allbad = 0, []
    for path in paths:
        if not os.path.exists(path) or not os.path.exists(OBJDUMP):
            print(f"cannot inspect {path}", file=sys.stderr)
            return 2
        cos = device_code_objects(path)
        if not cos:
            print(f"no gfx950 code object in {path}", file=sys.stderr)
            return 2
        for co in cos:
            with tempfile.NamedTemporaryFile(suffix=".co") as tf:
                tf.write(co)
                tf.flush()
                dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", tf.name], capture_output=True, text=True).stdout
            seen, bad = lint(dis)
            total += seen
            allbad += bad
    for b in allbad:
        print("VIOLATION:", b)
    print(f"{total} wide vector-memory stores checked, {len(allbad)} violation(s)")
    return 1 if allbad else 0


if __name__ == "__main__":
    sys.exit(main())
