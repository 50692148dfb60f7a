"""CPU: the boundary is a C ABI -- compile a plain-C consumer of include/valle_engine.h with gcc, link it against
libvalle_engine.so and run it (host-side quantiser, error reporting without a GPU)."""
import os
import shutil
import subprocess

import pytest

import valle_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_plain_c_program_links_and_runs(tmp_path):
    lib = valle_amd._lib.LIB_PATH
    assert os.path.isfile(lib), "build first: python __graft_entry__.py build"
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(lib)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
           "-o", exe, "-L", libdir, "-lvalle_engine", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi_smoke ok" in r.stdout
