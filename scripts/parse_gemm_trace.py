"""usage: parse_gemm_trace.py <prof_dir>   (prof_dir holds rocprofv3's *kernel_trace.csv and order.json)"""
import csv
import glob
import json
import sys

d = sys.argv[1]
order = json.loads(open(f"{d}/order.json").readline())
rows = [r for r in csv.DictReader(open(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]))
        if "gemm_" in r["Kernel_Name"] and "split_bf16x3" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split-K launches add a reduce kernel: group consecutive kernels per call by counting main kernels
main = [r for r in rows if "splitk_reduce" not in r["Kernel_Name"]]
i = 0
for o in order:
    durs = []
    for _ in range(o["reps"]):
        durs.append(int(main[i]["End_Timestamp"]) - int(main[i]["Start_Timestamp"]))
        i += 1
    us = sorted(durs)[len(durs) // 2] / 1e3
    flops = 2.0 * o["m"] * o["n"] * o["k"]
    mult = {"bf16x3": 6, "f16x2": 3, "f16x2p": 3}.get(o["kernel"], 1)
    print(f"{o['tag']:14s} {o['kernel']:7s} M={o['m']:6d} N={o['n']:5d} K={o['k']:5d}  {us:8.1f} us  "
          f"{flops / us / 1e6:7.1f} TFLOP/s fp32-equiv  {mult * flops / us / 1e6:7.1f} TFLOP/s MFMA-issued")
