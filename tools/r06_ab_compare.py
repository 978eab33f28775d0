#!/usr/bin/env python3
"""Compare the outputs of tools/r06_sweep_ab.sh: product build against a variant (same box)."""
import json, sys
d = sys.argv[1]; v = sys.argv[2] if len(sys.argv) > 2 else "r06base"
a = json.load(open(f"{d}/sweep_product.json")); b = json.load(open(f"{d}/sweep_{v}.json"))
ta = tb = 0
for x, y in zip(a, b):
    print(x["frame"], x["n"], "rounds", x["rounds"], y["rounds"], "us", x["sweep_us"], y["sweep_us"], "events", x["events"], y["events"])
    ta += x["sweep_us"]; tb += y["sweep_us"]
print("sweep total us: product", ta, v, tb)
a = json.load(open(f"{d}/arrange_product.json")); b = json.load(open(f"{d}/arrange_{v}.json"))
for x, y in zip(a, b):
    print(x["beam"], x["frame"], x["n"], "rounds", x["rounds"], y["rounds"], "us", x["sweep_us"], y["sweep_us"], "call ms", x["arrange_call_ms"], y["arrange_call_ms"])
print(open(f"{d}/sweep_product_phases.txt").read().splitlines()[0])
print(open(f"{d}/sweep_{v}_phases.txt").read().splitlines()[0])
