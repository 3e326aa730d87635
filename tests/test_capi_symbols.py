"""The C-ABI library loads without a GPU and exports every function include/rt_abi.h declares;
the ctypes mirror covers all of them.  No compute calls here."""
import ctypes as C
import os
import re

from raytracer_amd import capi

HEADER = os.path.join(capi.REPO_ROOT, "include", "rt_abi.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in rt_abi.h but not exported"


def test_ctypes_mirror_covers_the_header():
    assert sorted(capi.PROTOTYPES) == declared_functions()


def test_struct_sizes_match_the_shader_layouts():
    assert C.sizeof(capi.RtSphere) == 32 and C.sizeof(capi.RtTriangle) == 64 and C.sizeof(capi.RtVertex) == 16
    assert C.sizeof(capi.RtMaterial) == 80 and C.sizeof(capi.RtLight) == 48 and C.sizeof(capi.RtMeshInfo) == 12
    assert C.sizeof(capi.RtCamera) == 336 and C.sizeof(capi.RtPushConstants) == 44
    assert capi.RtCamera.viewInv.offset == 64 and capi.RtCamera.position.offset == 320 and capi.RtCamera.tanHalfFOV.offset == 332
    assert capi.RtPushConstants.reset_frame_accumulation.offset == 28 and capi.RtPushConstants.total_meshes.offset == 40
    assert capi.RtTriangle.normal.offset == 48 and capi.RtTriangle.mat.offset == 60


def test_defaults_and_errors_without_gpu():
    lib = capi.load()
    p = capi.RtParams()
    lib.rt_default_params(C.byref(p))
    assert (p.spp, p.max_bounces, p.accel) == (5, 20, capi.RT_ACCEL_BVH)  # raytracer.comp:117-118
    assert lib.rt_abi_version() == capi.RT_ABI_VERSION
    assert lib.rt_stats_size() == C.sizeof(capi.RtStats)
    assert lib.rt_render_frame(None, None, None, None) == capi.RT_ERR_INVALID_ARGUMENT
    assert lib.rt_scene_create(99, None, C.byref(C.c_void_p())) == capi.RT_ERR_INVALID_ARGUMENT


def test_ctypes_structs_have_the_headers_sizes(tmp_path):
    """Every struct the library reads or fills through a pointer must have the size the C compiler gives it from
    include/rt_abi.h: a field added on one side only would make the library write past a Python-side buffer."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "a C compiler is part of the build environment"
    names = ["RtSphere", "RtMaterial", "RtLight", "RtTriangle", "RtVertex", "RtMeshInfo", "RtCamera", "RtPushConstants",
             "RtParams", "RtStats", "RtSceneView", "RtControllerState"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rt_abi.h"\nint main(void) {\n' +
                   "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run([cc, "-std=c99", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n in names:
        assert C.sizeof(getattr(capi, n)) == int(out[n]), f"{n}: ctypes {C.sizeof(getattr(capi, n))} bytes, header {out[n]}"


def test_a_c_host_sees_the_librarys_abi_version_and_stats_size(tmp_path):
    """A C host compiled against include/rt_abi.h and linked with the library: the header's RT_ABI_VERSION and its
    sizeof(RtStats) must be what the library was built with (rt_get_stats writes rt_stats_size() bytes into the
    caller's struct), and the ctypes mirror must carry the same version."""
    import re
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    assert cc, "a C compiler is part of the build environment"
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include "rt_abi.h"\nint main(void) {\n'
                   '  printf("%d %d %zu %zu\\n", RT_ABI_VERSION, rt_abi_version(), sizeof(RtStats), rt_stats_size());\n'
                   '  return (RT_ABI_VERSION == rt_abi_version() && sizeof(RtStats) == rt_stats_size()) ? 0 : 1;\n}\n')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.run([cc, "-std=c99", "-I", os.path.dirname(HEADER), str(src), "-L", libdir, "-lraytracer_amd",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    header_version = int(re.search(r"#define RT_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert header_version == capi.RT_ABI_VERSION == int(r.stdout.split()[0])
