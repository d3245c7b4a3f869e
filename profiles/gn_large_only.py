"""The `gn_large` line of bench.py alone (one estimate_transform on 64M device-generated pairs, past
the Infinity Cache), for rocprofv3 runs whose per-kernel averages are not diluted by 1M-point launches:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ktl -- python3 profiles/gn_large_only.py
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcl_fetch -- python3 profiles/gn_large_only.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcl_write -- python3 profiles/gn_large_only.py
    python3 profiles/collect_traffic.py gpurun_out/pmcl_fetch gpurun_out/pmcl_write profiles/r01_traffic_pmc_gn_large.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    import icp_rust_amd as I

    I.build()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 1024 * 1024
    print(json.dumps(bench.gn_large(n)))
