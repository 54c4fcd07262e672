"""Multi-dimensional blocks: 8 x 8 tiles of a 2-D array, 4 x 4 x 4 tiles of a 3-D array, separable
orthonormal DCT (include/dctz_hip.h, dctz_amd/csrc/dct_nd_block.h).  The reference's library has no such path (it
flattens every array, dctz-test.c:77-91; the hint is its FFTW r2r experiment, dct-fftw-test.c:74-97), so this mode's
parity is pinned on the DEFINITION only: scipy.fft.dctn / idctn(norm="ortho") -- "parity unpinned" by the reference.
CPU part: the product's block transform (run on the CPU by tests/emu) == the oracle's pinned flow bit for bit, both ==
scipy to rounding, the oracle's two flows agree to rounding, gather/scatter round trip, error bound of the codec.
GPU part (-m gpu, through the C ABI): every stream and the reconstruction bit-identical to the oracle."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
from scipy.fft import dctn, idctn

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "emu", "emu_dct64.so")
SHAPES = {2: (8, 8), 3: (4, 4, 4)}


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "emu", "emu_dct64.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-o", SO, src])
    return C.CDLL(SO)


def field(shape, dtype, seed=5, noise=0.01, amp=37.0):
    rng = np.random.default_rng(seed)
    axes = [np.linspace(0, 1, d) for d in shape]
    g = np.meshgrid(*axes, indexing="ij")
    f = np.sin(5 * np.pi * g[0]) * np.cos(3 * np.pi * g[1])
    if len(shape) == 3:
        f = f * np.sin(2 * np.pi * g[2] + 0.3) + 0.2 * np.sin(9 * np.pi * g[0] * g[1] * g[2])
    return (amp * (f + noise * rng.standard_normal(shape))).astype(dtype)


@pytest.mark.parametrize("nd", [2, 3])
@pytest.mark.parametrize("dtype,suf,tol", [(np.float64, "f64", 4e-14), (np.float32, "f32", 2e-5)])
def test_block_transform_product_oracle_scipy(emu, nd, dtype, suf, tol):
    rng = np.random.default_rng(17 + nd)
    fn = getattr(emu, "emu_nd_" + suf)
    for i in range(400):
        a = (rng.standard_normal(64) * 10 ** rng.uniform(-2, 1.5)).astype(dtype)
        if i == 0:
            a[:] = 1
        for inverse in (0, 1):
            b = np.empty_like(a)
            fn(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), nd - 1, inverse)
            of = (O.dct_inv if inverse else O.dct_fwd)(a, O.geom_impl(nd, O.FAST))
            on = (O.dct_inv if inverse else O.dct_fwd)(a, O.geom_impl(nd, O.NAIVE))
            assert np.array_equal(b.view(np.uint8), of.view(np.uint8)), "product's lane code != oracle's pinned flow"
            ref = (idctn if inverse else dctn)(a.astype(np.float64).reshape(SHAPES[nd]), type=2, norm="ortho").ravel()
            scale = np.abs(a).max() * 8
            assert np.abs(of - ref).max() <= tol * scale and np.abs(on - ref).max() <= tol * scale


@pytest.mark.parametrize("shape", [(8, 8), (45, 77), (1, 9), (16, 8), (4, 4, 4), (13, 22, 35), (1, 1, 5), (8, 12, 4)])
def test_gather_scatter_round_trip_and_padding(shape):
    x = np.arange(int(np.prod(shape)), dtype=np.float64).reshape(shape)
    lin = O.nd_gather(x)
    e = O.GEOM_EDGE[len(shape)]
    assert lin.size == 64 * int(np.prod([(d + e - 1) // e for d in shape]))
    assert np.array_equal(O.nd_scatter(lin, shape), x)
    assert lin.max() == x.max() and lin.min() == x.min()          # padding repeats samples


@pytest.mark.parametrize("shape,dtype", [((45, 77), np.float64), ((64, 96), np.float32), ((13, 22, 35), np.float64), ((16, 16, 32), np.float32)])
@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_oracle_nd_codec_bound_and_flows(shape, dtype, mode):
    x = field(shape, dtype)
    eb = 1e-3
    c = O.compress_nd(x, eb, mode, O.FAST)
    nblk = c.dc.size
    assert c.bin_index.size == nblk * 64 and np.all(c.bin_index[::64] == 255)
    assert int((c.bin_index == 255).sum()) == c.cnt + nblk
    r = O.decompress_nd(c, shape, O.FAST)
    scaled = (x / x.dtype.type(c.sf)) if c.sf != 1 else x
    err = np.abs(r.astype(np.float64) - scaled.astype(np.float64) * c.sf).max()
    trunc = 2.0 ** -24 * 8.0 * 10.0 * (1 if mode == O.EC else 1 + 26.5 * 8)
    assert err <= (np.sqrt(63.0) * eb * 1.07 + trunc) * c.sf * (1.0 if dtype == np.float64 else 1.5) + (0 if dtype == np.float64 else 2e-6 * 37 * 64)
    # the definition-order flow agrees to rounding: same exceptions up to edge cases, PSNR equal
    c2 = O.compress_nd(x, eb, mode, O.NAIVE)
    differ = int((c.bin_index != c2.bin_index).sum())
    assert differ <= (0 if dtype == np.float64 else max(2, c.bin_index.size // 2000)), differ
    r2 = O.decompress_nd(c2, shape, O.NAIVE)
    assert abs(O.psnr(scaled.ravel() * x.dtype.type(c.sf), r.ravel())["psnr"] - O.psnr(scaled.ravel() * x.dtype.type(c.sf), r2.ravel())["psnr"]) < 1e-3


def test_nd_blocks_beat_flat_blocks_on_smooth_fields():
    """Why the mode exists: on a smooth 2-D / 3-D field a tile has far fewer out-of-range coefficients than a 64-element
    run along the fastest axis (recorded, not a parity statement)."""
    x2 = field((256, 256), np.float32, noise=0.0)
    x3 = field((64, 64, 64), np.float64, noise=0.0)
    for x in (x2, x3):
        flat = O.compress(x.ravel(), 1e-3, O.EC)
        tiled = O.compress_nd(x, 1e-3, O.EC)
        assert tiled.cnt < flat.cnt


# ------------------------------------------------------------------------------------------------- GPU --
@pytest.fixture(scope="module")
def ctx():
    import dctz_amd
    c = dctz_amd.Context(0)
    yield c
    c.close()


# ragged shapes go through the gather / scatter passes; shapes whose extents are multiples of the tile edge are read and
# written in place by the big kernels (NdDirect: partial last tiles, one block per row, many tiles per workgroup)
CASES = [((45, 77), np.float64), ((360, 720), np.float32), ((64, 8), np.float64), ((13, 22, 35), np.float64),
         ((64, 64, 64), np.float32), ((4, 4, 260), np.float64), ((100, 7, 9), np.float32), ((64, 128), np.float64),
         ((8, 16, 64), np.float64), ((16, 8), np.float32), ((256, 1544), np.float64), ((40, 36, 52), np.float64),
         ((4, 4, 4), np.float32)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", CASES)
@pytest.mark.parametrize("mode", [O.EC, O.QT])
def test_hip_nd_streams_bit_exact(ctx, shape, dtype, mode):
    import torch
    import dctz_amd
    x = field(shape, dtype, seed=11 + len(shape))
    eb = 1e-3
    c = O.compress_nd(x, eb, mode, O.FAST)
    hmode = dctz_amd.QT if mode == O.QT else dctz_amd.EC
    xd = torch.from_numpy(x).to(ctx.device)
    scaled = torch.empty_like(xd)
    out, info = ctx.compress_nd(xd, eb, hmode, scaled=scaled)
    assert info.sf == c.sf and info.cnt == c.cnt and info.nblk == c.dc.size
    assert info.max_abs == c.stats.max and info.min_abs == c.stats.min
    # device-order sum: the reference's serial float sum of a near-zero-mean field is mostly
    # rounding; compare on the scale of the data
    assert abs(info.mean - c.mean) <= 1e-5 * c.stats.max
    assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
    assert np.array_equal(out["dc"].cpu().numpy().view(np.uint32), c.dc.view(np.uint32))
    assert np.array_equal(out["ac_exact"][:c.cnt].cpu().numpy().view(np.uint32), c.ac_exact.view(np.uint32))
    want_scaled = (x / x.dtype.type(c.sf)) if c.sf != 1 else x
    assert np.array_equal(scaled.cpu().numpy(), want_scaled)
    if mode == O.QT:
        assert np.array_equal(np.array(info.qtable[1:], dtype=dtype), c.qtable[1:])
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    r = ctx.decompress_nd(out, info.cnt, shape, tdt, eb, info.sf, hmode, qtable=np.array(info.qtable[:])).cpu().numpy()
    assert np.array_equal(r, O.decompress_nd(c, shape, O.FAST))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype,mode", [((1800, 3600), np.float32, O.EC), ((128, 256, 160), np.float64, O.QT)])
def test_hip_nd_in_place_large_equals_oracle_and_gather_path(shape, dtype, mode):
    """Large enough for the speculative single pass: the array is read in place (no gather), statistics fused into
    k_compress behind the device-chosen scaling factor; streams == oracle, and == what the gather / scatter path gives."""
    import torch
    import dctz_amd
    x = field(shape, dtype, seed=31, noise=0.002)
    c = O.compress_nd(x, 1e-3, mode, O.FAST)
    hmode = dctz_amd.QT if mode == O.QT else dctz_amd.EC
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    res = {}
    for direct in ("1", "0"):
        os.environ["DCTZHIP_ND_DIRECT"] = direct
        try:
            cx = dctz_amd.Context(0)
        finally:
            os.environ.pop("DCTZHIP_ND_DIRECT", None)
        xd = torch.from_numpy(x).to(cx.device)
        out, info = cx.compress_nd(xd, 1e-3, hmode)
        assert info.sf == c.sf and info.cnt == c.cnt
        if direct == "1" and os.environ.get("DCTZHIP_SPECULATE", "1") != "0":
            assert info.flags & dctz_amd.hip.INFO_STATS_FUSED
        assert np.array_equal(out["bin_index"].cpu().numpy(), c.bin_index)
        assert np.array_equal(out["dc"].cpu().numpy().view(np.uint32), c.dc.view(np.uint32))
        assert np.array_equal(out["ac_exact"][:c.cnt].cpu().numpy().view(np.uint32), c.ac_exact.view(np.uint32))
        r = cx.decompress_nd(out, info.cnt, shape, tdt, 1e-3, info.sf, hmode, qtable=np.array(info.qtable[:])).cpu().numpy()
        res[direct] = r
        cx.close()
    assert np.array_equal(res["1"], res["0"]) and np.array_equal(res["1"], O.decompress_nd(c, shape, O.FAST))


@pytest.mark.gpu
def test_hip_nd_against_scipy_definition(ctx):
    """Independent of the oracle: with a bound far below the data's smallest coefficient spacing nothing is binned
    differently by rounding noise, and the reconstruction must match idctn(quantised dctn) -- here simply: the error
    stays inside the codec's bound and the PSNR equals the oracle's definition-order flow."""
    import torch
    import dctz_amd
    x = field((96, 160), np.float64, seed=3)
    out, info = ctx.compress_nd(torch.from_numpy(x).to(ctx.device), 1e-4, dctz_amd.EC)
    r = ctx.decompress_nd(out, info.cnt, x.shape, torch.float64, 1e-4, info.sf, dctz_amd.EC).cpu().numpy()
    cn = O.compress_nd(x, 1e-4, O.EC, O.NAIVE)
    rn = O.decompress_nd(cn, x.shape, O.NAIVE)
    assert info.cnt == cn.cnt
    assert np.abs(r - x).max() <= (np.sqrt(63.0) * 1e-4 * 1.07 + 2.0 ** -24 * 80) * info.sf
    assert abs(O.psnr(x.ravel(), r.ravel())["psnr"] - O.psnr(x.ravel(), rn.ravel())["psnr"]) < 1e-6


@pytest.mark.gpu
def test_hip_nd_rejects_bad_shapes(ctx):
    import torch
    import dctz_amd
    with pytest.raises(dctz_amd.hip.DctzHipError):
        ctx.nd_blocks((5,))
    with pytest.raises(dctz_amd.hip.DctzHipError):
        ctx.nd_blocks((4, 0, 4))


# ------------------------------------------------------------------------- drop-in library, container, CLI --
ROOT = os.path.dirname(HERE)
LIBDIR = os.path.join(ROOT, "dctz_amd", "lib")
BIN = os.path.join(ROOT, "dctz_amd", "bin")


def _nd_container(x, eb, mode):
    """A multi-dimensional .z assembled from the oracle's streams: the reference's layout (dctz-comp-lib.c:775-820)
    with the geometry in bits 8..15 of `datatype` and "DZND" + three extents behind the last section (include/dctz.h)."""
    import struct
    import zlib
    c = O.compress_nd(x, eb, mode, O.FAST)
    z = [zlib.compress(c.bin_index.tobytes()), zlib.compress(c.dc.tobytes()), zlib.compress(c.ac_exact.tobytes())]
    is_d = x.dtype == np.float64
    h = bytearray(56)
    struct.pack_into("<II", h, 0, (1 if is_d else 0) | (x.ndim << 8), x.size)
    struct.pack_into("<d", h, 8, eb)
    struct.pack_into("<I", h, 16, c.cnt)
    struct.pack_into("<d" if is_d else "<f", h, 24, c.sf)
    struct.pack_into("<d" if is_d else "<f", h, 32, c.mean)
    struct.pack_into("<III", h, 40, len(z[0]), len(z[1]), len(z[2]))
    if mode == O.QT:
        struct.pack_into("<I", h, 52, c.bin_index.size)
    blob = bytes(h) + b"".join(z)
    if mode == O.QT:
        blob += c.qtable.tobytes()
    dims = list(x.shape) + [0] * (3 - x.ndim)
    blob += struct.pack("<IIII", 0x444E5A44, *dims)
    return blob, c


@pytest.mark.parametrize("mode,variant", [(O.EC, "ec"), (O.QT, "qt")])
def test_nd_container_check_and_dump(tmp_path, mode, variant):
    """Host only: dctz_check_container (shallow + deep) and dctz-dump -v on oracle-built multi-dimensional containers."""
    if not os.path.exists(os.path.join(BIN, "dctz-dump")):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(os.path.join(LIBDIR, f"libdctz-{variant}.so"))
    lib.dctz_check_container.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    for x in (field((45, 77), np.float64), field((13, 22, 35), np.float32)):
        blob, c = _nd_container(x, 1e-3, mode)
        buf = np.frombuffer(blob, np.uint8).copy()
        assert lib.dctz_check_container(buf.ctypes.data, buf.size, 0, 1) == 0
        assert lib.dctz_check_container(buf.ctypes.data, buf.size - 1, 0, 0) == -1           # truncated
        bad = buf.copy(); bad[-4] ^= 1                                                        # an extent that no longer multiplies to N
        assert lib.dctz_check_container(bad.ctypes.data, bad.size, 0, 0) == -2
        f = tmp_path / f"nd{x.ndim}.z"
        f.write_bytes(blob)
        r = subprocess.run([os.path.join(BIN, "dctz-dump"), "-v", str(f)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
        assert f"N={x.size}" in r.stdout and "multi-dimensional blocks: " + " x ".join(str(d) for d in x.shape) in r.stdout
        assert f"({c.bin_index.size} raw)" in r.stdout and "= layout" in r.stdout


class _TVarBuf(C.Union):
    _fields_ = [("f", C.POINTER(C.c_float)), ("d", C.POINTER(C.c_double))]


class _TVar(C.Structure):   # dctz.h:49-59
    _fields_ = [("datatype", C.c_int), ("err_bound", C.c_double), ("var_name", C.c_char_p), ("buf", _TVarBuf)]


def _tvar(arr):
    v = _TVar()
    v.datatype = 1 if arr.dtype == np.float64 else 0
    if arr.dtype == np.float64:
        v.buf.d = arr.ctypes.data_as(C.POINTER(C.c_double))
    else:
        v.buf.f = arr.ctypes.data_as(C.POINTER(C.c_float))
    return v


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["ec", "qt"])
@pytest.mark.parametrize("shape,dtype", [((90, 130), np.float32), ((13, 22, 35), np.float64)])
@pytest.mark.parametrize("how", ["call", "env"])
def test_dropin_nd_container_equals_oracle(variant, shape, dtype, how):
    """dctz_set_block_dims / DCTZ_BLOCK_DIMS + dctz_compress: the container is byte for byte the one assembled from the
    oracle's streams (same zlib), the caller's buffer holds x / sf, dctz_decompress gives the oracle's reconstruction,
    and the next call is flat again."""
    os.environ["DCTZ_QUIET"] = "1"
    lib = C.CDLL(os.path.join(LIBDIR, f"libdctz-{variant}.so"))
    lib.dctz_compress.argtypes = [C.POINTER(_TVar), C.c_int, C.POINTER(C.c_size_t), C.POINTER(_TVar), C.c_double]
    lib.dctz_decompress.argtypes = [C.POINTER(_TVar), C.POINTER(_TVar)]
    lib.dctz_set_block_dims.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
    mode = O.QT if variant == "qt" else O.EC
    x = field(shape, dtype, seed=23)
    want, c = _nd_container(x, 1e-3, mode)
    work = x.copy()
    n = x.size
    zbuf = np.zeros(n * x.itemsize + 65536, np.uint8)
    rec = np.zeros(n, dtype)
    var, var_z, var_r = _tvar(work.reshape(-1)), _TVar(), _tvar(rec)
    var_z.datatype = var.datatype
    var_z.buf.d = zbuf.ctypes.data_as(C.POINTER(C.c_double))
    out_size = C.c_size_t(0)
    try:
        if how == "call":
            assert lib.dctz_set_block_dims(len(shape), (C.c_size_t * len(shape))(*shape)) == 0
        else:
            os.environ["DCTZ_BLOCK_DIMS"] = "x".join(str(d) for d in shape)
        assert lib.dctz_compress(C.byref(var), n, C.byref(out_size), C.byref(var_z), 1e-3) == 1
    finally:
        os.environ.pop("DCTZ_BLOCK_DIMS", None)
    got = bytes(zbuf[:out_size.value])
    assert len(got) == len(want)
    assert got[:32] == want[:32] and got[40:] == want[40:]            # everything but the header's mean (below)
    mean = np.frombuffer(got[32:40], dtype)[0]
    assert mean == dtype(c.mean)                                     # serial-order mean of the original array: bit-exact
    scaled = (x / dtype(c.sf)) if c.sf != 1 else x
    assert np.array_equal(work, scaled)
    assert lib.dctz_decompress(C.byref(var_z), C.byref(var_r)) == 1
    assert np.array_equal(rec.reshape(shape), O.decompress_nd(c, shape, O.FAST))
    # the request was for one call: the same array again is compressed flat
    work2 = x.copy().reshape(-1)
    var2 = _tvar(work2)
    assert lib.dctz_compress(C.byref(var2), n, C.byref(out_size), C.byref(var_z), 1e-3) == 1
    assert (int(np.frombuffer(bytes(zbuf[:4]), np.uint32)[0]) >> 8) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["ec", "qt"])
def test_dropin_nd_container_with_the_entropy_stage_on_the_gpu(variant):
    """DCTZ_ZLIB_GPU=1 on a multi-dimensional call: sections inflate (zlib) to the oracle's streams, both trailers are
    there ("DZND" extents, then the "DZIX" chunk index), dctz_check_container accepts it, dctz_decompress -- which takes
    the indexed path -- returns the oracle's reconstruction."""
    import struct
    import zlib
    os.environ["DCTZ_QUIET"] = "1"
    lib = C.CDLL(os.path.join(LIBDIR, f"libdctz-{variant}.so"))
    lib.dctz_compress.argtypes = [C.POINTER(_TVar), C.c_int, C.POINTER(C.c_size_t), C.POINTER(_TVar), C.c_double]
    lib.dctz_decompress.argtypes = [C.POINTER(_TVar), C.POINTER(_TVar)]
    lib.dctz_set_block_dims.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
    lib.dctz_check_container.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    mode = O.QT if variant == "qt" else O.EC
    shape, dtype = (61, 44, 52), np.float64
    x = field(shape, dtype, seed=29)
    c = O.compress_nd(x, 1e-3, mode, O.FAST)
    work = x.copy()
    n = x.size
    zbuf = np.zeros(n * x.itemsize + 65536, np.uint8)
    rec = np.zeros(n, dtype)
    var, var_z, var_r = _tvar(work.reshape(-1)), _TVar(), _tvar(rec)
    var_z.datatype = var.datatype
    var_z.buf.d = zbuf.ctypes.data_as(C.POINTER(C.c_double))
    out_size = C.c_size_t(0)
    os.environ["DCTZ_ZLIB_GPU"] = "1"
    try:
        assert lib.dctz_set_block_dims(3, (C.c_size_t * 3)(*shape)) == 0
        assert lib.dctz_compress(C.byref(var), n, C.byref(out_size), C.byref(var_z), 1e-3) == 1
    finally:
        os.environ.pop("DCTZ_ZLIB_GPU", None)
    z = bytes(zbuf[:out_size.value])
    s0, s1, s2 = struct.unpack_from("<III", z, 40)
    off = 56
    for sz, want in zip((s0, s1, s2), (c.bin_index, c.dc, c.ac_exact)):
        assert z[off:off + 2] == b"\x78\x5e" and zlib.decompress(z[off:off + sz]) == want.tobytes()
        off += sz
    if mode == O.QT:
        off += 64 * 8
    assert struct.unpack_from("<IIII", z, off) == (0x444E5A44, *shape)
    magic, chunk, n0, n1, n2 = struct.unpack_from("<IIIII", z, off + 16)
    assert (magic, chunk) == (0x58495A44, 16384) and n0 == (c.bin_index.size + 16383) // 16384
    assert len(z) == off + 16 + ((20 + 2 * (n0 + n1 + n2) + 3) & ~3)
    assert lib.dctz_check_container(zbuf.ctypes.data, out_size.value, 0, 1) == 0
    assert lib.dctz_decompress(C.byref(var_z), C.byref(var_r)) == 1
    assert np.array_equal(rec.reshape(shape), O.decompress_nd(c, shape, O.FAST))
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".z") as f:
        f.write(z); f.flush()
        r = subprocess.run([os.path.join(BIN, "dctz-dump"), "-v", f.name], capture_output=True, text=True)
    assert r.returncode == 0 and "chunk index:" in r.stdout and "= layout" in r.stdout and "61 x 44 x 52" in r.stdout, r.stdout
    # the index trailer is part of what dctz_check_container vouches for: cut short or inconsistent -> refused
    assert lib.dctz_check_container(zbuf.ctypes.data, out_size.value - 8, 0, 0) == -1            # truncated inside the index
    zi = zbuf[:out_size.value].copy()
    zi[off + 16 + 20] ^= 0x01                                 # first chunk's size: the sizes no longer tile the stream
    assert lib.dctz_check_container(zi.ctypes.data, zi.size, 0, 0) < 0
    # a damaged chunk is detected by the indexed reader (adler32 of the content, as inflate() checks it), which then hands
    # the sections to the one-stream inflate: damage is treated as the reference's reader treats it (return code ignored,
    # dctz-decomp-lib.c:244-322) -- the call returns
    code = f"""
import ctypes as C, numpy as np, os, sys
sys.path.insert(0, {ROOT!r})
from tests.test_nd_blocks import _TVar, _tvar
os.environ["DCTZ_QUIET"] = "1"
lib = C.CDLL({os.path.join(LIBDIR, f"libdctz-{variant}.so")!r})
lib.dctz_decompress.argtypes = [C.POINTER(_TVar), C.POINTER(_TVar)]
z = np.fromfile(sys.argv[1], np.uint8)
rec = np.zeros({n}, np.float64)
vz, vr = _TVar(), _tvar(rec)
vz.datatype = 1
vz.buf.d = z.ctypes.data_as(C.POINTER(C.c_double))
lib.dctz_decompress(C.byref(vz), C.byref(vr))
print("returned")
"""
    import tempfile
    bad = bytearray(z)
    bad[56 + 2 + 40] ^= 0x10                                 # inside the first chunk of bin_index
    with tempfile.NamedTemporaryFile(suffix=".z") as f:
        f.write(bytes(bad)); f.flush()
        for dev in ("0", "1"):                                # host threads; the device decoder (which hands damage on to them)
            r = subprocess.run([sys.executable, "-c", code, f.name], capture_output=True, text=True, env=dict(os.environ, DCTZ_INFLATE_GPU=dev))
            assert r.returncode == 0 and "returned" in r.stdout and "does not inflate" in r.stderr, (r.returncode, r.stderr[-300:])


@pytest.mark.gpu
def test_cli_nd_blocks(tmp_path):
    """dctz-ec-test with DCTZ_ND_BLOCKS: the reference's argv (fastest extent first) -> tiles; CR improves on a smooth field."""
    x = field((180, 360), np.float32, seed=2, noise=0.0)
    src = tmp_path / "f.dat"
    x.tofile(src)
    env = dict(os.environ, DCTZ_QUIET="1")
    outs = {}
    for nd in (False, True):
        e = dict(env, DCTZ_ND_BLOCKS="1") if nd else env
        r = subprocess.run([os.path.join(BIN, "dctz-ec-test"), "-f", "1E-3", "var", str(src), "360", "180"], capture_output=True,
                           text=True, env=e, cwd=tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ("multi-dimensional blocks: 8 x 8 tiles" in r.stdout) == nd
        cr = float(r.stdout.split("CR = ")[1].split(",")[0]); psnr = float(r.stdout.split("PSNR = ")[1].split()[0])
        outs[nd] = (cr, psnr)
        rec = np.fromfile(str(src) + ".ec.1E-3.z.r", np.float32).reshape(x.shape)
        assert np.abs(rec - x).max() <= np.sqrt(63.0) * 1e-3 * 1.1 * 100 + 1e-3
    assert outs[True][0] > outs[False][0], outs
    d = subprocess.run([os.path.join(BIN, "dctz-dump"), "-v", str(src) + ".ec.1E-3.z"], capture_output=True, text=True)
    assert "multi-dimensional blocks: 180 x 360 array, 8 x 8 tiles" in d.stdout and "= layout" in d.stdout
