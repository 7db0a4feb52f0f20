import json,sys
d=json.loads(open(sys.argv[1]).read())
print(d["value"], d["ms_per_step"], "mlp frac", d["roofline"]["frac"], "stale", d["roofline"].get("traffic_stale"), "hbm frac", d["roofline_hbm_kernels"]["frac"], d["roofline_hbm_kernels"].get("traffic_stale"))
print({k:round(v["ms_per_step"],3) for k,v in d["workloads"].items()})
