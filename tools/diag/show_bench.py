import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","parity_ok","dtype")})
print(d["step_ms_by_gemm_form"])
print({k:(round(v["ms"],4), round(v.get("frac",0),3)) for k,v in d["stages"].items()})
print(d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"]["uniform_c8"]["frac"])
print(json.dumps(d.get("roofline_dominant"))[:1200])
print({k:v.get("parity",{}).get("ok") for k,v in d["configs"].items()}, str(d["self_check"])[:300])
