"""GPU entropy stage (dctz_amd/csrc/dctz_deflate.hip, include/dctz_hip.h: dctzhip_deflate).

Checker: zlib's own inflate -- what the reference's reader runs on every section (dctz-decomp-lib.c:244-322:
inflateInit + one inflate per section).  The device stream must inflate to the input exactly, for every size around the
chunk and segment boundaries, for incompressible, constant and DCTZ-shaped sections; and it must equal, byte for byte, the
host twin of the same routines (tests/emu/emu_deflate.cpp)."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

from oracle import oracle as O
from tests import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHUNK = 128 * 128


def twin():
    so = os.path.join(ROOT, "tests", "emu", "emu_deflate.so")
    src = os.path.join(ROOT, "tests", "emu", "emu_deflate.cpp")
    hdr = os.path.join(ROOT, "dctz_amd", "csrc", "deflate_chunk.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "dctz_amd", "csrc"), src, "-o", so], check=True)
    L = C.CDLL(so)
    L.emu_deflate.restype = C.c_size_t
    L.emu_deflate.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    L.emu_deflate_index.restype = C.c_size_t
    L.emu_deflate_index.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.emu_deflate_literals.restype = C.c_size_t
    L.emu_deflate_literals.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.emu_deflate_bound.restype = C.c_size_t
    L.emu_deflate_bound.argtypes = [C.c_size_t, C.c_int]
    return L


def twin_deflate(L, data, nthr=128, want_index=False, literals=False):
    a = np.frombuffer(data, dtype=np.uint8)
    cap = L.emu_deflate_bound(len(data), nthr)
    out = np.zeros(cap, dtype=np.uint8)
    chunk = nthr * 128
    sizes = np.zeros(max(1, (len(data) + chunk - 1) // chunk), np.uint32)
    n = (L.emu_deflate_literals if literals else L.emu_deflate_index)(a.ctypes.data if len(data) else None, len(data), out.ctypes.data, cap, nthr,
                                                                     sizes.ctypes.data)
    assert n > 0
    if want_index:
        return out[:n].tobytes(), sizes[:(len(data) + chunk - 1) // chunk]
    return out[:n].tobytes()


def inflate_by_index(z, sizes, n, chunk=CHUNK):
    """What a reader with the chunk index does: every chunk is a raw deflate stream of its own."""
    assert z[:2] == b"\x78\x5e"
    out, off = [], 2
    for j, sz in enumerate(sizes):
        d = zlib.decompressobj(-15)
        piece = d.decompress(z[off:off + int(sz)])
        assert d.unconsumed_tail == b"" and d.unused_data == b"" and not d.eof      # all of it read, no final block inside
        assert len(piece) == min(chunk, n - j * chunk)
        out.append(piece)
        off += int(sz)
    assert z[off:off + 2] == b"\x03\x00" and off + 6 == len(z)
    data = b"".join(out)
    assert int.from_bytes(z[-4:], "big") == zlib.adler32(data)
    return data


def sections():
    rng = np.random.default_rng(20261004)
    out = {"empty": b"", "one": b"\x07", "two": b"ab", "three_same": b"zzz"}
    for n in (127, 128, 129, 255, 256, 257, CHUNK - 1, CHUNK, CHUNK + 1, 3 * CHUNK + 130):
        out[f"skew{n}"] = rng.choice([127, 128, 126, 255, 3], p=[.8, .1, .05, .03, .02], size=n).astype(np.uint8).tobytes()
        out[f"rand{n}"] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    out["zeros"] = bytes(5 * CHUNK + 77)
    out["const"] = b"\x7f" * (2 * CHUNK)
    out["period64"] = bytes(range(64)) * 1000
    out["floats"] = rng.standard_normal(70001).astype(np.float32).tobytes()
    out["all_symbols"] = (bytes(range(256)) * 300)[:CHUNK * 4 + 5]
    # a distribution that drives code lengths past 15 bits before the repair (Fibonacci-like counts)
    fib, a, b = [], 1, 1
    for s in range(24):
        fib.append(bytes([s]) * a)
        a, b = b, a + b
    deep = b"".join(fib)
    out["deep_tree"] = bytes(rng.permutation(np.frombuffer(deep[:CHUNK], dtype=np.uint8)))
    return out


# ---------------------------------------------------------------- CPU: format of the twin --
@pytest.mark.parametrize("nthr", [128, 256])
def test_twin_streams_inflate_to_the_input(nthr):
    L = twin()
    for name, data in sections().items():
        z = twin_deflate(L, data, nthr)
        assert zlib.decompress(z) == data, name
        d = zlib.decompressobj()
        assert d.decompress(z) == data and d.eof and d.unused_data == b"", name      # one complete stream, nothing after it
        assert len(z) <= L.emu_deflate_bound(len(data), nthr)
        z2, sizes = twin_deflate(L, data, nthr, want_index=True)
        assert z2 == z and inflate_by_index(z, sizes, len(data), nthr * 128) == data, name     # chunks inflate on their own


def test_twin_on_dctz_streams_stays_close_to_zlib():
    """Container size with the device's method against zlib level 6 (the reference's Z_DEFAULT_COMPRESSION) on the
    streams of a compress call: within 3 % in total."""
    L = twin()
    x = W.c3(96)
    c = O.compress(x.ravel(), 1e-3, O.EC)
    ours = ref = 0
    for arr in (c.bin_index, c.dc, c.ac_exact):
        b = np.ascontiguousarray(arr).tobytes()
        z = twin_deflate(L, b)
        assert zlib.decompress(z) == b
        ours += len(z)
        ref += len(zlib.compress(b, 6))
    assert ours <= 1.03 * ref, (ours, ref)


# ---------------------------------------------------------------- GPU --
@pytest.fixture(scope="module")
def ctx():
    import dctz_amd
    c = dctz_amd.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_device_streams_inflate_and_equal_the_twin(ctx):
    import torch
    L = twin()
    cases = sections()
    names = list(cases)
    for i in range(0, len(names), 8):                      # up to 8 sections per call
        part = names[i:i + 8]
        dev = [torch.from_numpy(np.frombuffer(cases[k], dtype=np.uint8).copy()).to(ctx.device) for k in part]
        outs, index = ctx.deflate(dev, want_index=True)
        for k, o, ix in zip(part, outs, index):
            z = o.cpu().numpy().tobytes()
            assert zlib.decompress(z) == cases[k], k
            zt, ixt = twin_deflate(L, cases[k], want_index=True)
            assert z == zt and np.array_equal(ix, ixt), k
            assert inflate_by_index(z, ix, len(cases[k])) == cases[k], k


@pytest.mark.gpu
def test_misaligned_source(ctx):
    import torch
    rng = np.random.default_rng(5)
    base = rng.choice([1, 2, 3, 200], p=[.7, .1, .1, .1], size=3 * CHUNK + 11).astype(np.uint8)
    t = torch.from_numpy(base).to(ctx.device)
    for sh in (1, 2, 3):
        z = ctx.deflate([t[sh:]])[0].cpu().numpy().tobytes()
        assert zlib.decompress(z) == base[sh:].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_streams_of_a_compress_call(ctx, mode):
    """bin_index / DC / AC_exact as k_compress leaves them in HBM -> three zlib streams; inflated they are the oracle's
    streams; the total stays within 3 % of zlib level 6."""
    import torch
    x = W.c3(128)
    xd = torch.from_numpy(x.ravel().copy()).to(ctx.device)
    out, info = ctx.compress(xd, 1e-3, mode)
    cnt = int(info.cnt)
    secs = [out["bin_index"], out["dc"], out["ac_exact"][:cnt]]
    zs = [o.cpu().numpy().tobytes() for o in ctx.deflate(secs)]
    c = O.compress(x.ravel(), 1e-3, mode)
    ours = ref = 0
    for z, want in zip(zs, (c.bin_index, c.dc, c.ac_exact)):
        assert zlib.decompress(z) == np.ascontiguousarray(want).tobytes()
        ours += len(z)
        ref += len(zlib.compress(np.ascontiguousarray(want).tobytes(), 6))
    assert ours <= 1.03 * ref, (ours, ref)


@pytest.mark.gpu
def test_random_sections_against_the_twin(ctx):
    """64 sections of random length and statistics (constant stretches, periodic patterns, skewed and uniform bytes mixed
    in random proportions): device bytes == twin bytes, and zlib inflates them to the input."""
    import torch
    L = twin()
    rng = np.random.default_rng(77)
    cases = []
    for _ in range(64):
        n = int(rng.integers(1, 5 * CHUNK))
        parts, left = [], n
        while left > 0:
            m = int(min(left, rng.integers(1, 3000)))
            kind = rng.integers(0, 5)
            if kind == 0:
                parts.append(np.full(m, rng.integers(0, 256), np.uint8))
            elif kind == 1:
                per = rng.integers(0, 256, int(rng.choice([2, 4, 8, 16, 64, 128, 7, 63])), dtype=np.uint8)
                parts.append(np.resize(per, m))
            elif kind == 2:
                parts.append(rng.choice([127, 128, 126, 255, 0, 9], p=[.6, .2, .1, .05, .03, .02], size=m).astype(np.uint8))
            elif kind == 3:
                parts.append(rng.integers(0, 256, m, dtype=np.uint8))
            else:
                parts.append(rng.integers(0, 4, m, dtype=np.uint8))
            left -= m
        cases.append(np.concatenate(parts).tobytes())
    for i in range(0, len(cases), 8):
        part = cases[i:i + 8]
        dev = [torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(ctx.device) for b in part]
        outs, index = ctx.deflate(dev, want_index=True)
        for b, o, ix in zip(part, outs, index):
            z = o.cpu().numpy().tobytes()
            assert zlib.decompress(z) == b
            assert z == twin_deflate(L, b)
            assert inflate_by_index(z, ix, len(b)) == b


@pytest.mark.gpu
def test_argument_checks(ctx):
    import torch
    import dctz_amd
    t = torch.zeros(1000, dtype=torch.uint8, device=ctx.device)
    small = torch.zeros(10, dtype=torch.uint8, device=ctx.device)
    src = (C.c_void_p * 1)(t.data_ptr()); dst = (C.c_void_p * 1)(small.data_ptr())
    n = (C.c_size_t * 1)(1000); cap = (C.c_size_t * 1)(10); ln = (C.c_size_t * 1)()
    assert ctx.lib.dctzhip_deflate(ctx.h, 1, src, n, dst, cap, ln, None) != 0          # output below dctzhip_deflate_bound
    assert ctx.lib.dctzhip_deflate(ctx.h, 9, src, n, dst, cap, ln, None) != 0          # more than 8 sections
    assert int(ctx.lib.dctzhip_deflate_bound(0)) == 8 and int(ctx.lib.dctzhip_deflate_chunk_bytes()) == CHUNK


# ---------------------------------------------------------------- inflate on the GPU --
@pytest.mark.gpu
def test_device_inflate_round_trip(ctx):
    """dctzhip_inflate on what dctzhip_deflate wrote: every section of the format tests and 24 random mixtures come back
    byte for byte, with ok = 1 (lengths and the adler32 of the content agree with the stream)."""
    import torch
    cases = list(sections().values())
    rng = np.random.default_rng(99)
    for _ in range(24):
        n = int(rng.integers(1, 6 * CHUNK))
        kind = rng.integers(0, 3)
        if kind == 0:
            b = rng.choice([127, 128, 126, 255, 0, 9], p=[.6, .2, .1, .05, .03, .02], size=n).astype(np.uint8)
        elif kind == 1:
            b = np.repeat(rng.integers(0, 256, n // 37 + 1, dtype=np.uint8), 37)[:n]
        else:
            b = rng.integers(0, 256, n, dtype=np.uint8)
        cases.append(b.tobytes())
    for i in range(0, len(cases), 8):
        part = cases[i:i + 8]
        dev = [torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(ctx.device) for b in part]
        zs, index = ctx.deflate(dev, want_index=True)
        outs, ok = ctx.inflate(zs, index, [len(b) for b in part])
        assert ok
        for b, o in zip(part, outs):
            assert o.cpu().numpy().tobytes() == b


@pytest.mark.gpu
def test_device_inflate_rejects_damage_without_touching_anything_else(ctx):
    """Single-byte damage anywhere in a stream (header of a block, code lengths, tokens, stored bytes, the frame) and a
    wrong index: the kernel stays inside its buffers (guard bytes behind the output untouched), and either reports
    ok = 0 or -- when the flipped bits were padding -- still returns the exact input."""
    import torch
    rng = np.random.default_rng(4242)
    data = (rng.choice([127, 128, 126, 255, 0, 9], p=[.6, .2, .1, .05, .03, .02], size=3 * CHUNK + 777).astype(np.uint8).tobytes()
            + rng.integers(0, 256, CHUNK, dtype=np.uint8).tobytes() + bytes(CHUNK // 2))
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(ctx.device)
    (z,), (ix,) = ctx.deflate([src], want_index=True)
    zh = z.cpu().numpy().copy()
    n = len(data)
    guard = 4096
    dst = torch.full((n + guard,), 0xA5, dtype=torch.uint8, device=ctx.device)

    def run(zbytes, index):
        zt = torch.from_numpy(zbytes).to(ctx.device)
        dst[n:] = 0xA5
        zp = (C.c_void_p * 1)(zt.data_ptr()); zl = (C.c_size_t * 1)(zt.numel())
        ixa = np.ascontiguousarray(index, dtype=np.uint32)
        ixp = (C.c_void_p * 1)(ixa.ctypes.data); raw = (C.c_size_t * 1)(n); dp = (C.c_void_p * 1)(dst.data_ptr())
        ok = C.c_int(0)
        assert ctx.lib.dctzhip_inflate(ctx.h, 1, zp, zl, ixp, raw, dp, C.byref(ok)) == 0
        torch.cuda.synchronize()
        assert bool((dst[n:] == 0xA5).all()), "wrote behind the output"
        return bool(ok.value), dst[:n].cpu().numpy().tobytes()

    ok, out = run(zh, ix)
    assert ok and out == data
    rejected = 0
    for pos in list(rng.integers(0, len(zh), 150)) + [0, 1, 2, 3, len(zh) - 1, len(zh) - 5, len(zh) - 6]:
        bad = zh.copy()
        bad[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
        ok, out = run(bad, ix)
        assert (not ok) or out == data
        rejected += not ok
    assert rejected > 100
    wrong = ix.copy(); wrong[0] += 1; wrong[1] -= 1                      # chunk boundaries moved by one byte
    ok, out = run(zh, wrong)
    assert not ok
    wrong = ix.copy(); wrong[-1] += 1                                    # the sizes no longer tile the stream
    ok, out = run(zh, wrong)
    assert not ok


def test_stream_format_matches_the_committed_digests():
    """The bytes of the format are pinned: a change of the candidate set, the chunk size, the tie rules of the tree or
    the header coding shows up here (containers written by an older build stay readable either way -- they are zlib
    streams -- but the twin / device comparison alone would not notice a silent change of both)."""
    import hashlib
    import json
    path = os.path.join(ROOT, "tests", "golden", "deflate_streams.json")
    L = twin()
    secs = sections()
    gold = json.load(open(path))
    if os.environ.get("DCTZ_REGEN_GOLDEN"):
        for k in gold["streams"]:
            z = twin_deflate(L, secs[k])
            gold["streams"][k] = {"n": len(secs[k]), "stream_bytes": len(z), "sha256": hashlib.sha256(z).hexdigest()}
        json.dump(gold, open(path, "w"), indent=1)
    for k, g in gold["streams"].items():
        z = twin_deflate(L, secs[k])
        assert (len(secs[k]), len(z), hashlib.sha256(z).hexdigest()) == (g["n"], g["stream_bytes"], g["sha256"]), k


@pytest.mark.gpu
def test_c_program_moves_the_zlib_tail_to_the_device(tmp_path):
    """tests/c/entropy_stage.c (the code INTEGRATION.md section B shows): gcc against include/dctz_hip.h + zlib; the
    sections made on the device inflate with zlib's uncompress() and decode to within the error bound."""
    exe = str(tmp_path / "entropy_stage")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", os.path.join(ROOT, "tests", "c", "entropy_stage.c"), "-I", os.path.join(ROOT, "include"),
                           "-L", os.path.join(ROOT, "dctz_amd", "lib"), "-ldctzhip", "-Wl,-rpath," + os.path.join(ROOT, "dctz_amd", "lib"), "-lz", "-lm", "-o", exe])
    n, eb = 64 * 9000 + 13, 1e-3
    r = subprocess.run([exe, str(n), str(eb)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    tag, n_out, cnt, raw, zbytes, err = r.stdout.split()
    assert tag == "ENTROPY" and int(n_out) == n and int(raw) == n + 4 * ((n + 63) // 64) + 4 * int(cnt) and int(zbytes) < int(raw)
    assert float(err) <= 8 * eb                          # sqrt(63) * eb on the scaled values


@pytest.mark.gpu
def test_literals_only_sections(ctx):
    """DCTZHIP_DEFLATE_LITERALS (what the drop-in sets for DC and AC_exact): no match search; device == twin byte for
    byte, zlib inflates it, and on bytes of floats the section is as small as with the search (0.2 %)."""
    import torch
    L = twin()
    rng = np.random.default_rng(31)
    floats = (rng.standard_normal(150001) * 3.0).astype(np.float32).tobytes()
    runs = bytes(3 * CHUNK + 5)
    dev = [torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(ctx.device) for b in (floats, runs, b"")]
    zs, index = ctx.deflate(dev, want_index=True, literals=[True, True, True])
    for b, z, ix in zip((floats, runs, b""), zs, index):
        zb = z.cpu().numpy().tobytes()
        zt, ixt = twin_deflate(L, b, want_index=True, literals=True)
        assert zb == zt and np.array_equal(ix, ixt) and zlib.decompress(zb) == b
    with_search = ctx.deflate(dev[:1])[0].numel()
    assert abs(zs[0].numel() - with_search) <= 0.002 * with_search
    back, ok = ctx.inflate(zs, index, [len(floats), len(runs), 0])
    assert ok and back[0].cpu().numpy().tobytes() == floats
