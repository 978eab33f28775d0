"""Cost of the Gaussian mixture selection stage (jamd_gms_apply_dev) beside the scoring it follows:
256 utterances x 300 frames, 2000-state model, 129 selection states x 16 Gaussians, -gsnum 24."""
import json
import time

import numpy as np

from julius_amd import lib, synth


def main():
    eng = lib.Engine(0)
    full = synth.make_gmm(S=2000, M=16, D=39, seed=1)
    gs = synth.make_gmm(S=129, M=16, D=39, seed=2)
    rng = np.random.default_rng(3)
    state2gs = rng.integers(0, 129, size=2000).astype(np.int32)
    nutt, tper = 256, 300
    T = nutt * tper
    fr = synth.make_frames(full, T=T, seed=4)
    off = (np.arange(nutt + 1) * tper).astype(np.int32)
    gm = lib.Gmm(eng, full)
    stage = lib.Gms(eng, gs, state2gs, 24)
    d_fr = lib.DevBuf(eng, fr.nbytes).upload(fr)
    d_sc = lib.DevBuf(eng, 4 * T * 2000)
    out = {}
    strict = lib.Gms(eng, gs, state2gs, 24).set_strict_order(True)
    for name, fn in (("scoring", lambda: gm.outprob_dev(d_fr.ptr, T, d_sc.ptr)),
                     ("selection", lambda: stage.apply_dev(d_fr.ptr, T, d_sc.ptr, off)),
                     ("selection_strict", lambda: strict.apply_dev(d_fr.ptr, T, d_sc.ptr, off))):
        fn(); eng.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        eng.sync()
        out[name + "_ms"] = (time.perf_counter() - t0) / 5 * 1e3
    out["frames"] = T
    out["selection_us_per_frame_per_utt"] = out["selection_ms"] * 1e3 / tper
    print(json.dumps(out))


if __name__ == "__main__":
    main()
