"""Which HIP / RCCL instances serve a process that loads librdamd before or after torch
(diagnostic for rdamd_comm_*).  usage: python profiles/rccl_probe.py [torch-first]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
import root_digger_amd as rd
if len(sys.argv) == 1:
    import torch
try:
    c = rd.Comm(rd.Comm.unique_id(), 0, 1)
    print("comm ok")
except Exception as e:
    print("comm FAILED:", e)
for line in open("/proc/self/maps"):
    if any(k in line for k in ("rccl", "amdhip64", "hsa-runtime")) and "r-xp" in line:
        print(line.split()[-1])
