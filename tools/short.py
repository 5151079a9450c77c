import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"): continue
    d = json.loads(line)
    print(d["config"]["pipeline"], d["value"], d["ms_per_step"], {k: (v["avg_kernel_us"], v["frac"]) for k, v in d["kernels"].items()})
