"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection CSVs).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly half of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section), so the read side is doubled.  Writes profiles-ready JSON.
usage: python tools/pmc_summary.py <fetch.csv> <write.csv> [out.json]"""
import collections
import csv
import json
import sys

csv.field_size_limit(1 << 30)


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
        key = "%s grid=%s" % (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        agg[key].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [0.0])
        w = write.get(k, [0.0])
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        out[k] = {"launches": len(f), "fetch_size_kib_raw": fk, "write_size_kib": wk,
                  "hbm_read_bytes_corrected": 2.0 * fk * 1024.0, "hbm_write_bytes": wk * 1024.0,
                  "hbm_bytes_per_launch": 2.0 * fk * 1024.0 + wk * 1024.0}
    top = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])
    for k, v in top[:25]:
        print("%-110s n=%4d  rd %9.2f MB  wr %9.2f MB" % (k, v["launches"], v["hbm_read_bytes_corrected"] / 1e6,
                                                          v["hbm_write_bytes"] / 1e6))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
