"""Prints mean counter values per (kernel, grid) from a rocprofv3 counter_collection CSV."""
import collections, csv, sys
csv.field_size_limit(1 << 30)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    key = "%s g=%d" % (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[key]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in agg.items():
    print(k)
    for n, v in sorted(c.items()):
        print("    %-32s %16.1f  (n=%d)" % (n, sum(v) / len(v), len(v)))
