import sys, json, torch
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(bench.fem_axle_entry("cuda:0"), indent=1))
print(json.dumps(bench.fem_axle_entry("cuda:0", tol_rate=1e-6), indent=1))
print(json.dumps(bench.fem_axle_entry("cuda:0", steps=4, streaming=True, tol_rate=1e-6), indent=1))
