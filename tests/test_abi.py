"""CPU suite: the C-ABI library loads without a GPU and exports every symbol include/*.h declares; argument
validation and error reporting work without touching a device."""
import ctypes as C
import os
import re

import pytest

import server_amd as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cghip_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header", ["cghip.h", "cghip_debug.h"])
def test_library_exports_every_declared_symbol(header):
    names = declared_functions(header)
    assert len(names) >= 3
    L = C.CDLL(A.library_paths()[0])
    for n in names:
        assert hasattr(L, n), "%s declared in include/%s but not exported by libcghip.so" % (n, header)


def test_header_declares_the_whole_boundary():
    names = declared_functions("cghip.h")
    for n in ("cghip_device_create", "cghip_device_info", "cghip_device_gc", "cghip_host_alloc",
              "cghip_plane_upload_async", "cghip_plane_peer_copy_async", "cghip_mixer_create",
              "cghip_mixer_render_async", "cghip_ticket_wait", "cghip_ticket_release", "cghip_last_error"):
        assert n in names


def test_headers_compile_as_c_and_cpp(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "cghip.h"\n#include "cghip_debug.h"\nint main(void){return sizeof(cghip_draw)>0?0:1;}\n')
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)
    subprocess.run(["g++", "-std=c++14", "-Wall", "-Werror", "-x", "c++", "-I", inc, "-c", str(src), "-o", str(tmp_path / "t2.o")], check=True)


def test_struct_layouts_match_ctypes():
    # sizes the Python binding assumes must equal the C compiler's
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write('#include <stdio.h>\n#include "cghip.h"\nint main(void){printf("%zu %zu %zu %zu %zu\\n",'
                             'sizeof(cghip_transform),sizeof(cghip_draw),sizeof(cghip_format),sizeof(cghip_surface),sizeof(cghip_coord));return 0;}\n')
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [C.sizeof(A.Transform), C.sizeof(A.Draw), C.sizeof(A.Format), C.sizeof(A.Surface), C.sizeof(A.Coord)]


def test_abi_version_and_error_string():
    L = A.lib()
    assert L.cghip_abi_version() == 1
    assert L.cghip_device_count(None) != 0
    assert b"null" in L.cghip_last_error()


def test_no_device_is_reported_not_crashed():
    if A.device_count() > 0:
        pytest.skip("a GPU is visible here")
    L = A.lib()
    dev = C.c_void_p()
    rc = L.cghip_device_create(0, C.byref(dev))
    assert rc == -2 and not dev.value  # CGHIP_ERR_NO_DEVICE, like the reference's not_supported (device.cpp:109-114)
    with pytest.raises(A.CGHIPError):
        A.Mixer(1, 0)


def test_product_never_uses_the_oracle():
    """The oracle is test infrastructure: no product source may import, include, link or load it."""
    pat = re.compile(r"(import\s+oracle|from\s+oracle|liboracle|oracle/|#include\s+\"[^\"]*ref_)")
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "server_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not pat.search(text), (dirpath, f)
