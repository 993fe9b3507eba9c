#!/bin/bash
# which ingredient makes the process abort at exit?  every case prints its exit status; LD_PRELOAD prints the native backtrace of the abort
mkdir -p gpurun_out/crash
export LD_PRELOAD=$PWD/tests/probes/abrt_bt.so
P="python3 -m pytest tests/test_gpu_batch.py -x -q -m gpu -p no:cacheprovider"
$P -k "copy_engine" > gpurun_out/crash/A_torch_only.log 2>&1; echo "A (torch test only) rc=$?"
$P -k "rccl" > gpurun_out/crash/B_rccl_only.log 2>&1; echo "B (rccl test only) rc=$?"
$P -k "rccl or copy_engine" > gpurun_out/crash/AB.log 2>&1; echo "AB rc=$?"
$P -k "not rccl and not copy_engine" > gpurun_out/crash/rest.log 2>&1; echo "rest rc=$?"
cat > /tmp/c.py <<'PY'
import sys, os
sys.path.insert(0, "lis-slam_amd")
import numpy as np
import lisreg
ctx = lisreg.Context(0)
ctx.close()
import torch
t = torch.zeros(1024).pin_memory()
print("C ok", torch.version.hip)
PY
python3 /tmp/c.py > gpurun_out/crash/C.log 2>&1; echo "C (lisreg then torch pin) rc=$?"
cat > /tmp/d.py <<'PY'
import sys
sys.path.insert(0, "lis-slam_amd")
import torch
t = torch.zeros(1024).pin_memory()
import lisreg
ctx = lisreg.Context(0)
ctx.close()
print("D ok")
PY
python3 /tmp/d.py > gpurun_out/crash/D.log 2>&1; echo "D (torch pin then lisreg) rc=$?"
cat > /tmp/e.py <<'PY'
import ctypes
h = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
n = ctypes.c_int(0); print("hipGetDeviceCount", h.hipGetDeviceCount(ctypes.byref(n)), n.value)
p = ctypes.c_void_p(); print("hipMalloc", h.hipMalloc(ctypes.byref(p), 1024)); h.hipFree(p)
import torch
t = torch.zeros(1024).pin_memory()
print("E ok")
PY
python3 /tmp/e.py > gpurun_out/crash/E.log 2>&1; echo "E (system hip then torch pin, no lisreg) rc=$?"
cat > /tmp/f.py <<'PY'
import ctypes
h = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
n = ctypes.c_int(0); print("hipGetDeviceCount", h.hipGetDeviceCount(ctypes.byref(n)), n.value)
r = ctypes.CDLL("/opt/rocm/lib/librccl.so", mode=ctypes.RTLD_GLOBAL)
import torch
t = torch.zeros(1024).pin_memory()
print("F ok")
PY
python3 /tmp/f.py > gpurun_out/crash/F.log 2>&1; echo "F (system hip + system rccl then torch pin) rc=$?"
for f in A_torch_only B_rccl_only AB C D E F; do echo "== $f"; grep -A40 "abrt_bt" gpurun_out/crash/$f.log | head -60; done
