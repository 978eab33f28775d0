"""The pin rests on oracle/refcfg/*/config.h (hand-written: the reference's own build system is not run to build
oracle/_ref).  This test runs the reference's OWN `./configure --enable-words-int` for libsent and libjulius in a
scratch copy and checks that the #define sets agree, except for the audio-input front end (OSS microphone, libfvad
VAD -- not part of the hot path and not compiled into oracle/_ref) and the build-description strings."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

REF = Path("/root/reference")
ROOT = Path(__file__).resolve().parent.parent
ALLOWED = {"AUDIO_API_DESC", "AUDIO_API_NAME", "HAS_OSS", "HAVE_SYS_SOUNDCARD_H", "USE_MIC", "HAVE_LIBFVAD",
           "JULIUS_BUILD_INFO", "JAMD_REFCFG_SENT_CONFIG_H", "JAMD_REFCFG_JULIUS_CONFIG_H"}


def _defines(path):
    out = {}
    for ln in Path(path).read_text(errors="replace").splitlines():
        m = re.match(r"#define\s+(\w+)\s*(.*)", ln)
        if m:
            out[m.group(1)] = m.group(2).strip()
    return out


@pytest.mark.skipif(not (REF / "libsent" / "configure").exists(), reason="/root/reference absent (GPU box)")
@pytest.mark.parametrize("lib,hdr,ours", [("libsent", "include/sent/config.h", "oracle/refcfg/sent/config.h"),
                                          ("libjulius", "include/julius/config.h", "oracle/refcfg/julius/config.h")])
def test_refcfg_matches_configure(tmp_path, lib, hdr, ours):
    shutil.copytree(REF / lib, tmp_path / lib)
    if (REF / "support").exists():
        shutil.copytree(REF / "support", tmp_path / "support")
    r = subprocess.run(["./configure", "--enable-words-int"], cwd=tmp_path / lib, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    theirs, mine = _defines(tmp_path / lib / hdr), _defines(ROOT / ours)
    diff = {k for k in set(theirs) | set(mine) if theirs.get(k) != mine.get(k)}
    assert diff <= ALLOWED, {k: (theirs.get(k), mine.get(k)) for k in diff - ALLOWED}
