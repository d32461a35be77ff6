"""bench JSON lines (a, b) -> per conv kernel launch time of both + whole-step ms.  usage: conv_probe_cmp.py a.json b.json"""
import json, sys
a, b = (json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:3])
print("step ms", a["ms_per_step"], b["ms_per_step"])
for k, v in a["conv_kernels"].items():
    o = b["conv_kernels"].get(k, {})
    print("%-42s n=%3s  %8.1f us -> %8.1f us   %6.3f -> %6.3f ms/step" % (k, v["launches_per_step"], v["avg_launch_us"], o.get("avg_launch_us", -1), v["ms_per_step"], o.get("ms_per_step", -1)))
