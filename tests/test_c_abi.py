"""The boundary is a genuine C-ABI: a C99 translation unit including only include/mpassit_amd.h compiles with gcc
(no C++), links against libmpassit_amd.so and -- on the GPU box -- runs without Python or torch in the process."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")


def _build(tmp_path):
    from mpassit_amd import build
    build.build()
    exe = str(tmp_path / "abi_smoke")
    lib = os.path.join(ROOT, "mpassit_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                    "-L" + lib, "-lmpassit_amd", "-lm", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_header_is_c99_and_links(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    try:
        import torch
        gpu = torch.cuda.is_available()
    except Exception:
        gpu = False
    if not gpu:   # no GPU here: the program must fail loudly at mpg_init, not fall back
        assert r.returncode != 0 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_program_runs_on_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_smoke ok" in r.stdout
