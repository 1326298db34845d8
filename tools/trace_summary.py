"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and, for one
steady-state step, the time line (start offset, duration, stream) so overlap and
idle gaps are visible.  usage: python tools/trace_summary.py <kernel_trace.csv> [n_steps]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        n = r["Kernel_Name"]
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        key = n.split("(")[0][:70] + " g=%sx%sx%s" % (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"])
        agg[key][0] += 1
        agg[key][1] += (r["e"] - r["s"]) / 1e3
    tot = sum(v[1] for v in agg.values())
    print("%-100s %6s %10s %9s" % ("kernel", "calls", "us/step", "avg us"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%-100s %6d %10.1f %9.1f" % (k, v[0], v[1] / nsteps, v[1] / v[0]))
    print("sum of kernel time per step: %.1f us" % (tot / nsteps))
    # last step time line: find the last clip_adam kernel and the one before it
    adam = [i for i, r in enumerate(rows) if "clip_adam" in r["Kernel_Name"]]
    if len(adam) >= 2 and "--timeline" in sys.argv:
        lo, hi = adam[-2] + 1, adam[-1] + 1
        t0 = rows[lo]["s"]
        busy_end = t0
        for r in rows[lo:hi]:
            gap = (r["s"] - busy_end) / 1e3
            busy_end = max(busy_end, r["e"])
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            print("%9.1f +%8.1f us  q%-3s gap %6.1f  %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, r["Queue_Id"], gap, n))
        print("step wall: %.1f us" % ((rows[hi - 1]["e"] - t0) / 1e3))


if __name__ == "__main__":
    main()
