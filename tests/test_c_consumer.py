"""The drop-in boundary used from plain C (tests/c_consumer/consumer.c): what a Rust `extern "C"`
binding does, with no Python in the process.  Compiling it with `gcc -std=c99 -pedantic` also checks
that include/icp_mi355x.h is a C header."""
import os
import shutil
import subprocess

import pytest

import icp_rust_amd as I

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "icp_rust_amd", "lib")


def build_consumer(tmp_path):
    I.build()
    exe = str(tmp_path / "consumer")
    rocm_lib = "/opt/rocm/lib"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_consumer", "consumer.c"), "-o", exe, "-L" + LIBDIR, "-licp_mi355x", "-lm",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + rocm_lib]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not found")
def test_header_is_c_and_the_library_links_from_c(tmp_path):
    exe = build_consumer(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    # without a GPU the program reports ICP_NO_DEVICE and exits 77 (no CPU fallback); with one it passes
    assert r.returncode in (0, 77), (r.returncode, r.stdout, r.stderr)
    if r.returncode == 77:
        assert "no HIP device" in r.stdout


@pytest.mark.gpu
def test_reference_3dscan_test_through_the_c_abi_from_plain_c(tmp_path):
    exe = build_consumer(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.startswith("ok:")
