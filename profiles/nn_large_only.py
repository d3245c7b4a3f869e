"""The `nn_large` line of bench.py alone (the search kernel on 16M x 16M device-generated clouds, past the Infinity
Cache), for rocprofv3 runs whose per-kernel averages are not diluted by 1M-point launches:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/knl -- python3 profiles/nn_large_only.py
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pnl_f -- python3 profiles/nn_large_only.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pnl_w -- python3 profiles/nn_large_only.py
    python3 profiles/collect_traffic.py gpurun_out/pnl_f gpurun_out/pnl_w profiles/r05_traffic_pmc_nn_large.json
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 1024 * 1024
    print(json.dumps(bench.nn_large(n)))
