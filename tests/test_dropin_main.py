"""The reference's own program (CR/main.cpp:3-18: construct a Renderer, set depth of field, RenderInParallel) must compile
and link against this library's Renderer (csrc/renderer.hpp) without a change to its text.  Compile and link only: running it
needs a GPU and the reference's scene_final.txt in the working directory.  The reference tree exists only where this build was
made, so the test is skipped everywhere else; nothing of the reference is copied (the compiler reads main.cpp where it lies)."""
import os
import shutil
import subprocess

import pytest

from helpers import ROOT

REF_MAIN = "/root/reference/CUDA_RayTracing/main.cpp"
CSRC = os.path.join(ROOT, "cuda_raytracing_amd", "csrc")
LIBDIR = os.path.join(ROOT, "cuda_raytracing_amd")


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference tree is not on this machine")
@pytest.mark.skipif(shutil.which("g++") is None, reason="no host C++ compiler")
def test_reference_main_compiles_and_links_against_the_renderer_shim(tmp_path):
    # main.cpp says #include "Renderer.h": one forwarding header stands in for the reference's (which pulls in OpenGL / ImGui)
    (tmp_path / "Renderer.h").write_text('#include "renderer.hpp"\n')
    exe = tmp_path / "main_dropin"
    # the source comes in on stdin so that the quoted include is looked up in -I order, not next to the reference's own Renderer.h
    cmd = ["g++", "-std=c++17", "-Wall", "-I" + str(tmp_path), "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-x", "c++", "-", "-x", "none",
           os.path.join(CSRC, "renderer.cpp"), "-L" + LIBDIR, "-lrt_amd", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)]
    with open(REF_MAIN, "rb") as src:
        r = subprocess.run(cmd, stdin=src, capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    syms = subprocess.run(["nm", "-C", str(exe)], capture_output=True, text=True).stdout
    for want in ("Renderer::Renderer(", "Renderer::RenderInParallel(", "rt_render_frame"):
        assert want in syms, want
