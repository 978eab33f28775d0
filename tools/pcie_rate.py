#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer scoring entry (jamd_gmm_outprob_host): host frames in, host [T][S] rows
out, with pageable buffers and with page-locked buffers from jamd_host_alloc().  DESIGN.md section 5."""
import ctypes as C, json, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from julius_amd import lib, synth

S, M, D, T = 3000, 16, 39, 16000
eng = lib.Engine(0)
model = synth.make_gmm(S=S, M=M, D=D, seed=0)
gm = lib.Gmm(eng, model)
frames = synth.make_frames(model, T=T, seed=1)
L = lib.load()

def rate(fr_ptr, out_ptr, n=5):
    L.jamd_gmm_outprob_host(gm.h, fr_ptr, T, out_ptr)
    t0 = time.perf_counter()
    for _ in range(n):
        assert L.jamd_gmm_outprob_host(gm.h, fr_ptr, T, out_ptr) == 0
    return (time.perf_counter() - t0) / n

out = np.empty((T, S), np.float32)
dt_pageable = rate(frames.ctypes.data, out.ctypes.data)
hp_in, hp_out = C.c_void_p(), C.c_void_p()
assert L.jamd_host_alloc(eng.h, frames.nbytes, C.byref(hp_in)) == 0 and L.jamd_host_alloc(eng.h, out.nbytes, C.byref(hp_out)) == 0
C.memmove(hp_in, frames.ctypes.data, frames.nbytes)
dt_pinned = rate(hp_in, hp_out)
pinned = np.ctypeslib.as_array(C.cast(hp_out, C.POINTER(C.c_float)), (T, S))
same = bool(np.array_equal(pinned, out))
L.jamd_host_free(eng.h, hp_in); L.jamd_host_free(eng.h, hp_out)
print(json.dumps({"frames": T, "states": S, "pageable_ms": dt_pageable * 1e3, "pageable_frame_states_per_s": T * S / dt_pageable,
                  "pinned_ms": dt_pinned * 1e3, "pinned_frame_states_per_s": T * S / dt_pinned,
                  "pinned_d2h_GBs": out.nbytes / dt_pinned / 1e9, "same_result": same}))
