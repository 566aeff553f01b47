"""One line per run of `bench.py --train-only`: step time and the attention / other class averages.  usage: c5_classes.py file.json [tag]"""
import json
import sys
d = json.load(open(sys.argv[1]))["c5_train_step"]
r = d.get("roofline_by_kernel", {})
print(sys.argv[2] if len(sys.argv) > 2 else "", round(d["ms_per_step"], 3), "ms |",
      " ".join(f"{k[6:]}={v.get('avg_launch_us') or v.get('ms_per_step')}" for k, v in r.items()))
