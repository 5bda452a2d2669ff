"""usage: parse_gemm_trace.py <prof_dir>   (prof_dir holds rocprofv3's *kernel_trace.csv and order.json)"""
import csv
import glob
import json
import sys

d = sys.argv[1]
order = json.loads(open(f"{d}/order.json").readline())
rows = [r for r in csv.DictReader(open(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]))
        if ("gemm_" in r["Kernel_Name"] or "splitk_reduce" in r["Kernel_Name"]) and "split_bf16x3" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split-K launches add a reduce kernel: a call = one main kernel + the reduce launches that follow it (round 6: their time is
# part of the call -- rounds 3-5 listed the main kernel alone)
calls = []
for r in rows:
    d_ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if "splitk_reduce" in r["Kernel_Name"]:
        if calls:
            calls[-1][1] += d_ns
    elif "row_split" in r["Kernel_Name"] or "row_amax" in r["Kernel_Name"]:
        continue
    else:
        calls.append([d_ns, 0])
i = 0
for o in order:
    durs = []
    for _ in range(o["reps"]):
        durs.append(calls[i][0] + calls[i][1])
        i += 1
    us = sorted(durs)[len(durs) // 2] / 1e3
    flops = 2.0 * o["m"] * o["n"] * o["k"]
    mult = {"bf16x3": 6, "f16x2": 3, "f16x2p": 3, "f16x2p8": 3, "narrow16": 3}.get(o["kernel"], 1)
    print(f"{o['tag']:14s} {o['kernel']:7s} M={o['m']:6d} N={o['n']:5d} K={o['k']:5d}  {us:8.1f} us  "
          f"{flops / us / 1e6:7.1f} TFLOP/s fp32-equiv  {mult * flops / us / 1e6:7.1f} TFLOP/s MFMA-issued")
