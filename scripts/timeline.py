"""Per-stream timeline of one steady-state env-step from a rocprofv3 kernel trace (kernel_trace.csv):
busy time per queue, idle gaps on the queue that runs the state pass, time per kernel family."""
import collections, csv, re, sys

path = sys.argv[1]
step_idx = int(sys.argv[2]) if len(sys.argv) > 2 else -3
rows = [r for r in csv.DictReader(open(path))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    n = n[5:] if n.startswith("void ") else n
    n = n.replace("lram::(anonymous namespace)::", "").replace("lram::", "")
    r["name"] = re.sub(r"\(.*$", "", n)[:48]
rows.sort(key=lambda r: r["s"])
# a step = from one action_argmax (last kernel of a step, second slice) to the next
arg = [i for i, r in enumerate(rows) if "action_argmax" in r["name"]]
# two slices -> two argmax per step; group ends in pairs
ends = arg[1::2][:int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 30]
lo, hi = ends[step_idx - 1] + 1, ends[step_idx] + 1
step = rows[lo:hi]
t0, t1 = step[0]["s"], max(r["e"] for r in step)
print(f"step of {len(step)} kernels, {(t1 - t0) / 1e6:.3f} ms")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r["e"] - r["s"] for r in rs)
    fam = collections.Counter()
    for r in rs:
        fam[r["name"]] += r["e"] - r["s"]
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.3f} ms  " + ", ".join(f"{k} {v / 1e6:.2f}" for k, v in fam.most_common(6)))
tot = collections.Counter()
cnt = collections.Counter()
for r in step:
    tot[r["name"]] += r["e"] - r["s"]
    cnt[r["name"]] += 1
print("--- kernel families (sum of durations, ms; count; avg us)")
for k, v in tot.most_common(20):
    print(f"{k:50s} {v / 1e6:7.3f} {cnt[k]:4d} {v / cnt[k] / 1e3:8.1f}")
# union busy time of the whole device & of state pass
def union(rs):
    iv = sorted((r["s"], r["e"]) for r in rs)
    tot_, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot_ += ce - cs
            cs, ce = s, e
    return tot_ + (ce - cs if cs is not None else 0)
cell = [r for r in step if "lazy_cell" in r["name"] or r["name"].startswith("mlstm_cell_kernel")]
print(f"device busy (union) {union(step) / 1e6:.3f} ms; state pass union {union(cell) / 1e6:.3f} ms")
if "-v" in sys.argv:
    for r in step:
        print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{r['Queue_Id']} {r['name']}")
