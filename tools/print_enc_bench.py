"""stdin: bench.py --only-encoder JSON; argv[1]: label.  One line: forward ms (without / with per-kernel events), TF, per-kernel ms."""
import json
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
try:
    e = json.loads(sys.stdin.read())["encode"]
    print(label, round(e["ms_per_batch"], 2), "ms (%.2f with events)" % e.get("ms_per_batch_with_kernel_events", 0.0),
          round(e["roofline"]["achieved"]), "TF", {k.replace("enc_", ""): round(v["ms_per_batch"], 2) for k, v in e["kernels"].items()})
except Exception as ex:  # noqa: BLE001
    print(label, "FAILED", ex)
