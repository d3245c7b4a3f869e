"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes) into the per-kernel traffic summary bench.py reads.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 40 --warmup 2 --brute-steps 0 --cpu-iters 0 --gn-points 0
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 40 --warmup 2 --brute-steps 0 --cpu-iters 0 --gn-points 0
    python profiles/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_traffic_pmc.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def read(dirname, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = (r.get("Dispatch_Id") or r.get("Correlation_Id"), r["Kernel_Name"])
            per_dispatch[key] += float(r["Counter_Value"])  # summed over XCDs / instances
            names[key] = r["Kernel_Name"]
        for key, v in per_dispatch.items():
            k = re.sub(r"\(.*", "", names[key].replace("(anonymous namespace)::", "")).replace("void ", "")
            acc[k][0] += v
            acc[k][1] += 1
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, write = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        e = {}
        if k in fetch:
            e["FETCH_SIZE_KB_avg_per_launch"] = round(fetch[k][0] / fetch[k][1], 1)
            e["launches_FETCH_SIZE"] = fetch[k][1]
        if k in write:
            e["WRITE_SIZE_KB_avg_per_launch"] = round(write[k][0] / write[k][1], 1)
            e["launches_WRITE_SIZE"] = write[k][1]
        kernels[k] = e
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, MI355X_MICROARCH.md "
            "'HBM' + 'rocprofv3 PMC slots'), python bench.py --steps 40 --warmup 2 --brute-steps 0 --cpu-iters 0 --gn-points 0, "
            "1M x 1M. Units: KB per launch. gfx950 correction: FETCH_SIZE reads exactly 1/2 of wide coalesced "
            "streaming reads -> to be doubled for the streaming GN kernels; k_nn_grid issues narrow per-lane "
            "gathers (uncalibrated width): reported uncorrected. Infinity-Cache hits are counted by these "
            "fabric-side counters, so at this 1M size (working set < 256 MiB) they are an upper bound on HBM bytes.")
    doc = {"note": note, "kernels": kernels}
    # one SEARCH = one launch of k_nn_grid_warm_coop (k_nn_grid_warm in rounds 2-3), of k_nn_grid_seeded (the first
    # search of an estimate call: seeds + the same walk) or of the general k_nn_grid; where the seeds are a launch of
    # their own (k_nn_grid_seed: certificates on) their bytes are spread over all searches
    tot = {"f": 0.0, "w": 0.0, "searches_f": 0, "searches_w": 0}
    for k, e in kernels.items():
        is_search = (k.startswith("icp::k_nn_grid<3, true") or k.startswith("icp::k_nn_grid_warm<3") or
                     k.startswith("icp::k_nn_grid_warm_coop<3") or k.startswith("icp::k_nn_grid_seeded<3"))
        if not (is_search or k.startswith("icp::k_nn_grid_seed<3")):
            continue
        tot["f"] += e.get("FETCH_SIZE_KB_avg_per_launch", 0.0) * e.get("launches_FETCH_SIZE", 0)
        tot["w"] += e.get("WRITE_SIZE_KB_avg_per_launch", 0.0) * e.get("launches_WRITE_SIZE", 0)
        if is_search:
            tot["searches_f"] += e.get("launches_FETCH_SIZE", 0)
            tot["searches_w"] += e.get("launches_WRITE_SIZE", 0)
    if tot["searches_f"] and tot["searches_w"]:
        fb, wb = tot["f"] / tot["searches_f"] * 1024, tot["w"] / tot["searches_w"] * 1024
        doc["k_nn_grid"] = {"traffic_bytes_per_launch": fb + wb, "fetch_bytes": fb, "write_bytes": wb,
                            "corrected": False, "launches": tot["searches_f"],
                            "weighting": "all search kernels (seed + warm + general) over the number of searches of the run"}
        for k, e in kernels.items():
            if (k.startswith("icp::k_nn_grid_warm<3") or k.startswith("icp::k_nn_grid_warm_coop<3")) and "FETCH_SIZE_KB_avg_per_launch" in e and "WRITE_SIZE_KB_avg_per_launch" in e:
                doc["k_nn_grid"]["warm_kernel_bytes_per_launch"] = 1024 * (e["FETCH_SIZE_KB_avg_per_launch"] +
                                                                            e["WRITE_SIZE_KB_avg_per_launch"])
            if k.startswith("icp::k_nn_grid_seed<3") and "FETCH_SIZE_KB_avg_per_launch" in e and "WRITE_SIZE_KB_avg_per_launch" in e:
                doc["k_nn_grid"]["seed_kernel_bytes_per_launch"] = 1024 * (e["FETCH_SIZE_KB_avg_per_launch"] +
                                                                            e["WRITE_SIZE_KB_avg_per_launch"])
    json.dump(doc, open(out, "w"), indent=1)
    for k, e in kernels.items():
        print(k, e)


if __name__ == "__main__":
    main()
