"""profiles/traffic.json from the PMC passes of scripts/profile_round.sh.
usage: make_traffic.py gpurun_out/prof_TAG KEY      (KEY as bench.py builds it, e.g. cfg3:16000000:overwrite:full:K)
       make_traffic.py --merge DIR/traffic_entry.json ...   (here, after gpurun merged gpurun_out/ back: profiles/ on
                                                             the GPU box is scratch, the entry files are what travels)
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, both reported in KiB: on gfx950 FETCH_SIZE tallies 128-B read
requests at 64 B (MI355X_MICROARCH.md, HBM section); scripts/calib/fetch_calib confirms the factor 2 for the 4, 8 and
16 B/lane loads these kernels issue and that WRITE_SIZE is exact (profiles/r01_fetch_calibration.txt)."""
import collections, csv, glob, hashlib, json, os, sys

def csrc_sha16():
    """hash of the kernel sources: bench.py reports a traffic figure only for the library it was measured with"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "falcon_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(root)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode()); h.update(open(os.path.join(root, fn), "rb").read())
    return h.hexdigest()[:16]

def mean_counter(d, name, kernel):
    v = [float(r["Counter_Value"]) for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(fn))
         if r["Counter_Name"] == name and kernel in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)

tf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
if sys.argv[1] == "--merge":
    d = json.load(open(tf)) if os.path.exists(tf) else {}
    for fn in sys.argv[2:]:
        e = json.load(open(fn))
        d[e["key"]] = e["traffic"]; d.setdefault("_detail", {})[e["key"]] = e["detail"]
        print("merged", e["key"], "%.3f GB" % (e["traffic"] / 1e9), e["detail"]["csrc_sha16"])
    json.dump(d, open(tf, "w"), indent=1, sort_keys=True)
    sys.exit(0)
base, key = sys.argv[1], sys.argv[2]
kernel = sys.argv[3] if len(sys.argv) > 3 else "assemble_"
fetch, nf = mean_counter(base + "/fetch", "FETCH_SIZE", kernel)
write, nw = mean_counter(base + "/write", "WRITE_SIZE", kernel)
traffic = (2.0 * fetch + write) * 1024.0
d = json.load(open(tf)) if os.path.exists(tf) else {}
d[key] = traffic
d.setdefault("_detail", {})[key] = {"FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB": write, "launches": [nf, nw],
                                    "read_bytes_corrected": 2 * fetch * 1024, "write_bytes": write * 1024,
                                    "source": os.path.relpath(base, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                                    "csrc_sha16": csrc_sha16()}
json.dump(d, open(tf, "w"), indent=1, sort_keys=True)
json.dump({"key": key, "traffic": traffic, "detail": d["_detail"][key]}, open(os.path.join(base, "traffic_entry.json"), "w"), indent=1)
print(key, "traffic %.3f GB (read %.3f, write %.3f)" % (traffic / 1e9, 2 * fetch * 1024 / 1e9, write * 1024 / 1e9))
