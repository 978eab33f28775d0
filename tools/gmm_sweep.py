#!/usr/bin/env python3
"""Time the D=39 GMM tile-kernel variants (JAMD_GMM_VARIANT) at the bench shape.
Each variant runs in a fresh process (the switch is read once)."""
import json, os, subprocess, sys
variants = sys.argv[1:] or ["0", "1", "2", "3", "4", "5"]
for v in variants:
    env = dict(os.environ, JAMD_GMM_VARIANT=v)
    r = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(v, f"{d['roofline']['kernel_ms']:.3f} ms", f"{d['value']:.3e} fs/s",
              f"valu {d['roofline']['valu']['achieved']:.1f} Tops/s", d["config"]["kernel"], "parity", d["parity_spot_check"], flush=True)
    except Exception as e:
        print(v, "FAILED", e, r.stderr[-500:])
