"""CPU-side checks of the boundary: every function include/*.h declares is
exported by the built libraries (no compute calls -- there is no GPU here), the
product never touches the oracle, and the product refuses to run without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dctz_amd", "lib")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not all(os.path.exists(os.path.join(LIB, f)) for f in ("libdctzhip.so", "libdctz-ec.so", "libdctz-qt.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "dctz_amd"), "all"])


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*?(?<!\\)$", "", src, flags=re.M)   # drop preprocessor lines (MAX/MIN macros)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined", "__typeof__")))


def _exported(so):
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(LIB, so)], text=True)
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def test_shim_exports_every_declared_symbol():
    decl = _declared("dctz_hip.h")
    assert len(decl) >= 20 and "dctzhip_compress" in decl and "dctzhip_decompress" in decl
    missing = [n for n in decl if n not in _exported("libdctzhip.so")]
    assert not missing, missing
    import dctz_amd
    dctz_amd.load_library()          # ctypes prototypes resolve too


@pytest.mark.parametrize("so", ["libdctz-ec.so", "libdctz-qt.so"])
def test_dropin_exports_reference_api(so):
    decl = _declared("dctz.h")
    for must in ("dctz_compress", "dctz_decompress", "calc_data_stat", "gen_bins", "gen_bins_f",
                 "compress_thread", "calc_psnr", "dct_init", "dct_fftw", "dct_fftw_f", "ifft_idct",
                 "ifft_idct_f", "dct_finish", "idct_finish", "dct_init_f", "dct_finish_f", "idct_finish_f"):
        assert must in decl, must                     # dctz.h:121-128, dct.h:17-27
    missing = [n for n in decl if n not in _exported(so)]
    assert not missing, missing
    C.CDLL(os.path.join(LIB, so))                     # loads, resolves libdctzhip via $ORIGIN


def test_header_layout_matches_reference_abi():
    """struct header is 56 bytes with the offsets recorded in SURVEY 8b; t_var is 32."""
    code = r'''
#include <stddef.h>
#include "dctz.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(struct header),
 offsetof(struct header,num_elements), offsetof(struct header,error_bound), offsetof(struct header,tot_AC_exact_count),
 offsetof(struct header,scaling_factor), offsetof(struct header,mean), offsetof(struct header,bindex_sz_compressed),
 offsetof(struct header,DC_sz_compressed), offsetof(struct header,AC_exact_sz_compressed), sizeof(t_var)); return 0; }
'''
    for flags, extra in (([], None), (["-DUSE_QTABLE"], 52)):
        exe = "/tmp/dctz_hdr_probe"
        subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe] + flags,
                       input=code, text=True, check=True)
        vals = [int(v) for v in subprocess.check_output([exe], text=True).split()]
        assert vals == [56, 4, 8, 16, 24, 32, 40, 44, 48, 32]


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dctz_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, f)
    for so in ("libdctzhip.so", "libdctz-ec.so", "libdctz-qt.so"):
        out = subprocess.check_output(["ldd", os.path.join(LIB, so)], text=True)
        assert "oracle" not in out


def test_no_cpu_fallback_without_gpu():
    import torch
    import dctz_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dctz_amd.DctzHipError):
        dctz_amd.Context(0)
    lib = dctz_amd.load_library()
    h = C.c_void_p()
    assert lib.dctzhip_ctx_create(C.byref(h), -1) != 0 and not h.value
    assert b"HIP" in lib.dctzhip_last_error(None) or b"device" in lib.dctzhip_last_error(None)


def test_rccl_datatype_values_match_the_header():
    """the shim calls the dlopen'ed RCCL with hand-written ncclDataType_t values (the header is not
    included: single-GPU users need no RCCL).  Compared with rccl.h's own text where the header is installed."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("rccl.h not installed")
    txt = open(hdr).read()
    src = open(os.path.join(ROOT, "dctz_amd", "csrc", "dctz_shim.hip")).read()
    vals = [re.search(r"#define DCTZ_NCCL_%s (\d+)" % k, src) for k in ("UINT8", "UINT64", "FLOAT32")]
    assert all(vals), "the shim's ncclDataType_t constants moved"
    assert "static_assert((int)ncclUint8 == DCTZ_NCCL_UINT8" in src     # (and the compiler checks them where the header is installed)
    for name, val in zip(("ncclUint8", "ncclUint64", "ncclFloat32"), [v.group(1) for v in vals]):
        h = re.search(name + r"\s*=\s*(\d+)", txt)
        assert h and h.group(1) == val, (name, val, h and h.group(1))


def test_an_rccl_of_another_major_version_is_refused():
    """The gather calls RCCL through dlsym'd pointers whose shapes (the by-value 128-byte id, the enum values) were read from
    the NCCL 2 header: a library that reports another major version is refused at load, with a message.
    The test double stands in for the library; no GPU is touched (the id call needs none)."""
    import subprocess
    double = os.path.join(ROOT, "tests", "c", "librccl_double.so")
    if not os.path.exists(double):
        pytest.skip("test double not built")
    code = ("import ctypes, sys; sys.path.insert(0, %r); from dctz_amd import hip as H; L = H.load_library(); "
            "b = ctypes.create_string_buffer(128); rc = L.dctzhip_comm_unique_id(b); "
            "print(rc, L.dctzhip_last_error(None).decode())" % ROOT)
    for ver, ok in (("22707", True), ("30100", False), ("1900", False)):
        env = dict(os.environ, DCTZHIP_RCCL_LIBRARY=double, RCCL_DOUBLE_VERSION=ver)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-1500:]
        rc, _, msg = r.stdout.strip().partition(" ")
        if ok:
            assert rc == "0", r.stdout
        else:
            assert rc != "0" and "NCCL 2 API" in msg, r.stdout
