"""The C++ host mirror of the reference's interfaces (lis-slam_amd/host/lis_slam_registration.hpp) above the C ABI: builds
with plain g++ and, on a GPU box, runs the reference's per-frame order end to end — VoxelGrid -> scan2SubMapOptimization ->
SubMapManager filters -> IterativeClosestPoint — checking the recovered pose inside the program."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "lis-slam_amd", "host")


def _build():
    import lisreg
    lisreg.lib()                                   # makes sure liblisreg.so exists (builds it if the tree is fresh)
    subprocess.check_call(["make", "-s", "-C", HOST])


def test_host_mirror_compiles():
    _build()
    assert os.path.exists(os.path.join(HOST, "host_smoke"))


@pytest.mark.gpu
def test_host_mirror_runs():
    _build()
    r = subprocess.run([os.path.join(HOST, "host_smoke")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "host_smoke ok" in r.stdout, r.stdout + r.stderr
