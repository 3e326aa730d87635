"""The example hosts on the C ABI (examples/): a plain C99 program and the C++ binding a maintainer
of the reference application would add (the stub the integration notes print, compiled).  Without a GPU they
are built and answer their no-argument self-checks; on the GPU the C program's image must be the
image the Python path renders, and the C++ program must find application-owned scene vectors and
the library's scene object rendering the same picture."""
import os
import re
import subprocess

import numpy as np
import pytest

from raytracer_amd import capi

ROOT = capi.REPO_ROOT
EX = os.path.join(ROOT, "examples")


def run(args, **kw):
    return subprocess.run(args, cwd=ROOT, capture_output=True, text=True, timeout=120, **kw)


def build_examples():
    r = subprocess.run(["make", "-C", EX], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r


def test_examples_build_against_the_header_and_library():
    r = build_examples()
    assert "warning" not in (r.stdout + r.stderr), r.stdout + r.stderr  # -Wall -Wextra (-pedantic for the C99 one)
    r = run([os.path.join(EX, "render_c")])
    assert r.returncode == 0 and "spp 5, max_bounces 20" in r.stdout, r.stdout + r.stderr
    r = run([os.path.join(EX, "hip_backend")])
    assert r.returncode == 0 and "layouts agree" in r.stdout, r.stdout + r.stderr


def test_integration_doc_shows_the_compiled_binding():
    """The HipBackend class printed in the integration notes is the one in examples/hip_backend.cpp
    (same calls in the same order), so the document cannot drift from code that compiles."""
    calls = lambda text: re.findall(r"\brt_[a-z0-9_]+(?=\()", text)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    doc_stub = doc[doc.index("struct HipBackend"):]
    doc_stub = doc_stub[:doc_stub.index("```")]
    src = open(os.path.join(EX, "hip_backend.cpp")).read()
    src_stub = src[src.index("struct HipBackend"):src.index("namespace {")]
    assert calls(doc_stub) == calls(src_stub)


@pytest.mark.gpu
def test_c_host_renders_what_the_python_path_renders(tmp_path):
    from helpers import gpu_frames
    from raytracer_amd.renderer import Context, Scene

    build_examples()  # gcc only, a no-op when build() has run
    out = tmp_path / "c.ppm"
    r = run([os.path.join(EX, "render_c"), "1", "64", "36", "2", "4", "2", str(out)])
    assert r.returncode == 0, r.stdout + r.stderr
    data = out.read_bytes()
    header = b"P6\n64 36\n255\n"
    assert data.startswith(header)
    rgb = np.frombuffer(data[len(header):], dtype=np.uint8).reshape(36, 64, 3)
    ctx = Context(0)
    g = gpu_frames(ctx, Scene(capi.RT_SCENE_CORNELL_STAR), 64, 36, spp=2, max_bounces=4, accel=capi.RT_ACCEL_BVH, frames=2)
    ctx.close()
    bgra = g["bgra"].reshape(36, 64, 4)
    assert np.array_equal(rgb, bgra[:, :, 2::-1])


@pytest.mark.gpu
def test_cpp_binding_with_application_owned_vectors():
    build_examples()
    r = run([os.path.join(EX, "hip_backend"), "run"])
    assert r.returncode == 0 and "identical" in r.stdout, r.stdout + r.stderr
