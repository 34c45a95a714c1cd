import sys,json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    for k,e in d.items():
        if not isinstance(e,dict): continue
        if "ms" in e: print(d["params"],"| single ms",e["ms"],"rounds",e.get("sssp_rounds"),"select",e.get("select_launch_ms"),"kern",e["stage_ms_per_cloud"].get("skeleton_kernels"), "parity", e.get("parity"))
        elif "ms_per_set" in e: print(d["params"],"| set ms",e["ms_per_set"],"select",e.get("select_launch_ms"),"kern/cloud",e["stage_ms_per_cloud"].get("skeleton_kernels"), "same", e.get("first_cloud_equals_its_single_call"))
