import sys,json
d=json.loads(sys.stdin.read())["encode"]; print(round(d["ms_per_batch"],3), {k:round(v["ms_per_batch"],3) for k,v in d["kernels"].items()})
