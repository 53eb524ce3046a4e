"""Known bytes / counter bytes for the kernels of scripts/ubench/fetch_size_gather.hip.
usage: pmc_gather_factor.py <known.json> <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> [out.json]
The counters are in KiB (rocprofv3 derived metrics); the LAST launch of every kernel is taken (the first warms the caches)."""
import csv
import json
import sys


def last_values(path, counter):
    vals = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].strip()
            if name.startswith("void "):
                name = name[5:]
            vals[name] = float(row["Counter_Value"])   # later rows overwrite earlier ones: the last dispatch stays
    return vals


def main():
    known = json.load(open(sys.argv[1]))
    fetch = last_values(sys.argv[2], "FETCH_SIZE")
    write = last_values(sys.argv[3], "WRITE_SIZE")
    out = {"unit": "bytes known / (counter x 1024)", "kernels": {}}
    for name, k in known["kernels"].items():
        f, w = fetch.get(name), write.get(name)
        out["kernels"][name] = {
            "read_bytes_known": k["read"], "FETCH_SIZE_KiB": f, "read_factor": round(k["read"] / (f * 1024), 3) if f else None,
            "written_bytes_known": k["written"], "WRITE_SIZE_KiB": w, "write_factor": round(k["written"] / (w * 1024), 3) if w else None,
            "table_MB": k.get("table_MB"), "note": k.get("note")}
    s = json.dumps(out, indent=1)
    print(s)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(s + "\n")


if __name__ == "__main__":
    main()
