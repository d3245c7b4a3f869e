# A/B of the snapshot re-ordering by cost class (ICP_RESORT_T: record batches / quads above which a query
# counts as expensive, 0 = off; ICP_RESORT_MODE 0: by the cold search's cost, 1: by the first warm search's)
mkdir -p gpurun_out/r2g
for M in 0 1; do for T in 0 4 6 8 12 16; do ICP_RESORT_MODE=$M ICP_RESORT_T=$T python3 bench.py --steps 100 --warmup 5 --brute-steps 0 --cpu-iters 0 --gn-points 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('mode $M T=$T', round(d['value'],1), 'it/s; search timed', round(d['roofline']['avg_launch_ms']*1e3,1), 'alone', round(d['roofline']['alone']['avg_launch_ms']*1e3,1), d['pose'][4])"; done; done
