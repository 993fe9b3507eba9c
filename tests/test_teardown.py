"""Process teardown: every way a host process can end with liblisreg.so in it must exit with status 0.

Round 3's GPU run passed all its tests and then aborted at interpreter exit (glibc "double free or corruption", status 134).  The
backtrace (tests/probes/abrt_bt.c, LD_PRELOAD) put the abort in the static destructor of librocm_smi64's std::map<DevInfoTypes, ...>:
lisreg_comm_* had dlopen'ed the SYSTEM librccl.so RTLD_GLOBAL (which brings the system librocm_smi64.so.1), a later test imported
torch, whose wheel carries its own librocm_smi64 (soname .so.7) — the global copy interposed the wheel's symbols and both destructors
freed the same map.  The same two lines reproduce it without liblisreg in the process at all.  Fixed on both sides: the library now
reuses an RCCL the process already has and otherwise loads one RTLD_LOCAL; the GPU tests no longer import torch.

Each case below is a fresh interpreter (subprocess), because the thing under test is its exit status.
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = f"""
import sys
sys.path.insert(0, {os.path.join(ROOT, "lis-slam_amd")!r})
import ctypes as C
import numpy as np
"""


def run_case(body: str, timeout: int = 300):
    code = PRELUDE + textwrap.dedent(body)
    env = dict(os.environ)
    env.setdefault("MALLOC_CHECK_", "3")                 # glibc aborts on the first inconsistent free, not on a later one
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=env)
    return r


def check(r, marker="CASE_DONE"):
    assert marker in r.stdout, f"the case did not reach its end:\n{r.stdout}\n{r.stderr}"
    assert r.returncode == 0, f"exit status {r.returncode}\n--- stdout\n{r.stdout}\n--- stderr\n{r.stderr[-3000:]}"


SMALL_BATCH = """
import lisreg
from lisreg import synth
cases = [synth.make_case(h=64, w=900, m_points=40000, scan_seed=3900 + i) for i in range(6)]
p = lisreg.default_params(1); p.fixed_iters = 2
n = len(cases)
T0 = np.stack([c["T_init"] for c in cases]).astype(np.float32)
def items():
    arr, keep = (lisreg.Item * n)(), []
    for i, c in enumerate(cases):
        sc = np.ascontiguousarray(c["src_corner"]); ss = np.ascontiguousarray(c["src_surf"]); keep += [sc, ss]
        arr[i].src_corner = sc.ctypes.data_as(C.c_void_p); arr[i].n_corner = len(sc)
        arr[i].src_surf = ss.ctypes.data_as(C.c_void_p); arr[i].n_surf = len(ss)
        arr[i].stride_bytes = sc.dtype.itemsize; arr[i].fmt = lisreg.FMT_XYZI
    return arr, keep
"""


@pytest.mark.gpu
def test_create_stage_run_destroy_exits_cleanly():
    """create -> stage_host_items (feeder threads, copy stream, pinned staging) -> prepare / run / fetch -> destroy -> exit."""
    check(run_case(SMALL_BATCH + """
ctx = lisreg.Context(0)
ctx.set_target(cases[0]["tgt_corner"], cases[0]["tgt_surf"])
arr, keep = items()
staged = (lisreg.Item * n)()
assert ctx._L.lisreg_stage_host_items(ctx._h, n, arr, staged) == 0
assert ctx._L.lisreg_batch_prepare(ctx._h, n, staged, C.byref(p), T0.ctypes.data_as(C.POINTER(C.c_float))) == 0
assert ctx._L.lisreg_batch_run(ctx._h) == 0
ctx._n_items = n
T, st = ctx.batch_fetch()
assert all(s["status"] == 0 for s in st)
ctx.close()
print("CASE_DONE")
"""))


@pytest.mark.gpu
def test_destroy_with_uploads_still_in_flight_exits_cleanly():
    """destroy right after staging: the copy stream still carries uploads out of the pinned buffers that are about to be freed."""
    check(run_case(SMALL_BATCH + """
for rep in range(3):
    ctx = lisreg.Context(0)
    ctx.set_target(cases[0]["tgt_corner"], cases[0]["tgt_surf"])
    arr, keep = items()
    staged = (lisreg.Item * n)()
    assert ctx._L.lisreg_stage_host_items(ctx._h, n, arr, staged) == 0
    assert ctx._L.lisreg_stage_host_items(ctx._h, n, arr, staged) == 0
    ctx.close()
print("CASE_DONE")
"""))


@pytest.mark.gpu
def test_leaked_context_exits_cleanly():
    """a context that is never destroyed (feeder threads parked on their condition variable, device memory live) at process exit."""
    check(run_case(SMALL_BATCH + """
ctx = lisreg.Context(0)
ctx.set_target(cases[0]["tgt_corner"], cases[0]["tgt_surf"])
arr, keep = items()
staged = (lisreg.Item * n)()
assert ctx._L.lisreg_stage_host_items(ctx._h, n, arr, staged) == 0
ctx._h = None            # leak it: no lisreg_destroy, neither now nor from __del__
leak2 = lisreg.Context(0)   # and one that reaches __del__ during interpreter shutdown
print("CASE_DONE")
"""))


@pytest.mark.gpu
def test_stage_error_paths_leave_no_live_job():
    """bad arguments after a good call, then a good call again: the thread pool and its tables survive refused calls."""
    check(run_case(SMALL_BATCH + """
ctx = lisreg.Context(0)
ctx.set_target(cases[0]["tgt_corner"], cases[0]["tgt_surf"])
arr, keep = items()
staged = (lisreg.Item * n)()
assert ctx._L.lisreg_stage_host_items(ctx._h, n, arr, staged) == 0
bad, _ = items(); bad[2].stride_bytes = 4
assert ctx._L.lisreg_stage_host_items(ctx._h, n, bad, staged) != 0
bad2, _ = items(); bad2[1].src_surf = None
assert ctx._L.lisreg_stage_host_items(ctx._h, n, bad2, staged) != 0
for _ in range(4):
    assert ctx._L.lisreg_stage_host_items(ctx._h, n, arr, staged) == 0
assert ctx._L.lisreg_batch_prepare(ctx._h, n, staged, C.byref(p), T0.ctypes.data_as(C.POINTER(C.c_float))) == 0
assert ctx._L.lisreg_batch_run(ctx._h) == 0
ctx._n_items = n
T, st = ctx.batch_fetch()
assert all(s["status"] == 0 for s in st)
ctx.close()
print("CASE_DONE")
"""))


@pytest.mark.gpu
def test_native_rccl_then_exit_is_clean():
    """lisreg_comm_* with one rank (librccl loaded by the library), context destroyed, exit."""
    check(run_case("""
import lisreg
ctx = lisreg.Context(0)
ctx.comm_init(0, 1, lisreg.comm_unique_id())
out = lisreg.DeviceArray(np.zeros((2, 12), np.float32))
src = lisreg.DeviceArray(np.arange(24, dtype=np.float32).reshape(2, 12))
ctx.gather_results(src.ptr, 2, out.ptr)
assert np.array_equal(lisreg.device_to_host(out.ptr, (2, 12)), np.arange(24, dtype=np.float32).reshape(2, 12))
ctx.close()
print("CASE_DONE")
"""))


def _have_torch():
    try:
        import importlib.util
        return importlib.util.find_spec("torch") is not None
    except Exception:
        return False


@pytest.mark.gpu
@pytest.mark.skipif(not _have_torch(), reason="torch not installed")
def test_native_rccl_then_torch_import_exits_cleanly():
    """The sequence that aborted round 3's GPU run at exit: the library's RCCL first, then `import torch` (a second set of ROCm
    libraries from the wheel) in the same process.  torch may or may not find the GPU behind a foreign HIP runtime — not our business —
    but the process must end with status 0."""
    check(run_case("""
import lisreg
ctx = lisreg.Context(0)
ctx.comm_init(0, 1, lisreg.comm_unique_id())
ctx.close()
import torch
try:
    t = torch.zeros(1024).pin_memory()
except Exception as e:
    print("torch behind the system HIP runtime:", type(e).__name__)
print("CASE_DONE")
"""))


@pytest.mark.gpu
@pytest.mark.skipif(not _have_torch(), reason="torch not installed")
def test_torch_first_then_native_rccl_exits_cleanly():
    """bench.py's order: torch (with its own HIP runtime and RCCL) first, liblisreg after it; lisreg_comm_* must pick up the RCCL that
    is already in the process (RTLD_NOLOAD) instead of loading the system's next to it."""
    r = run_case("""
import torch
torch.zeros(8, device="cuda:0").sum().item()
import lisreg
ctx = lisreg.Context(0)
ctx.comm_init(0, 1, lisreg.comm_unique_id())
out = lisreg.DeviceArray(np.zeros((2, 12), np.float32))
src = lisreg.DeviceArray(np.arange(24, dtype=np.float32).reshape(2, 12))
ctx.gather_results(src.ptr, 2, out.ptr)
assert np.array_equal(lisreg.device_to_host(out.ptr, (2, 12)), np.arange(24, dtype=np.float32).reshape(2, 12))
ctx.close()
maps = open("/proc/self/maps").read()
rccl = sorted({l.split()[-1] for l in maps.splitlines() if "librccl" in l})
print("rccl copies:", rccl)
assert len(rccl) == 1, rccl
print("CASE_DONE")
""")
    check(r)


def test_teardown_cases_need_a_gpu_and_say_so():
    """CPU box: lisreg_create refuses to run without a device (no fallback) and the interpreter still exits with status 0."""
    r = run_case("""
import lisreg
try:
    lisreg.Context(0)
    print("a device is visible")
except lisreg.LisregError as e:
    print("refused:", e)
print("CASE_DONE")
""")
    check(r)
