"""Per-queue view of the LAST lram_prefill in a rocprofv3 kernel trace of scripts/bench_prefill.py (kernel_trace.csv): busy time per
queue (= chunk lane), union busy time of the device, time per kernel family, how many kernels run concurrently."""
import collections, csv, re, sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    n = n[5:] if n.startswith("void ") else n
    n = n.replace("lram::(anonymous namespace)::", "").replace("lram::", "")
    r["name"] = re.sub(r"\(.*$", "", n)[:52]
rows.sort(key=lambda r: r["s"])
arg = [i for i, r in enumerate(rows) if "action_argmax" in r["name"]]
lo, hi = arg[-2] + 1, arg[-1] + 1
step = rows[lo:hi]
t0, t1 = step[0]["s"], max(r["e"] for r in step)
print(f"one prefill: {len(step)} kernels, {(t1 - t0) / 1e6:.1f} ms")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.1f} ms")
ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
depth, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[depth] += t - last
    last, depth = t, depth + d
print("time with k kernels in flight (ms): " + ", ".join(f"{k}: {v / 1e6:.1f}" for k, v in sorted(hist.items())))
tot, cnt = collections.Counter(), collections.Counter()
for r in step:
    tot[r["name"]] += r["e"] - r["s"]
    cnt[r["name"]] += 1
print("--- kernel families (sum of durations, ms; count; avg us)")
for k, v in tot.most_common(16):
    print(f"{k:54s} {v / 1e6:7.2f} {cnt[k]:5d} {v / cnt[k] / 1e3:8.1f}")
