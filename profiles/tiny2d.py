import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import icp_rust_amd as I
from icp_rust_amd.scans import load_scan2d
G = os.path.join(ROOT, 'tests', 'golden', 'scans2d')
src = load_scan2d(f'{G}/001.txt'); dst = load_scan2d(f'{G}/002.txt')
icp = I.Icp2d(dst)
for _ in range(5): icp.estimate(src, I.Transform(), 20)
