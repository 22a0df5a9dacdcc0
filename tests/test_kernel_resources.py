"""Build-time guard for the LDS-DMA kernels.  They wait with counted `s_waitcnt vmcnt(N)` ("all but my N youngest loads have
landed").  A register spill adds scratch loads/stores to the same counter, and scratch (flat-family) accesses return out of
order with buffer loads -- seen on the GPU as wrong weight rows at chunk boundaries when conv_patch3 once spilled 40 VGPRs.
So: none of those kernels may use scratch.  hipcc cross-compiles here; its resource-usage remarks are the evidence."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "climate2weather_amd", "csrc")
FILES = ["conv_igemm.hip", "conv_patch.hip", "conv_patch3.hip", "wgrad.hip", "wgrad_patch.hip"]


def _remarks(src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                          "--cuda-device-only", "-c", os.path.join(CSRC, src), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stderr


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_counted_vmcnt_kernels_do_not_spill():
    with ThreadPoolExecutor(max_workers=len(FILES)) as ex:
        logs = list(ex.map(_remarks, FILES))
    seen = 0
    for src, log in zip(FILES, logs):
        for blk in re.split(r"remark: Function Name: ", log)[1:]:
            name = blk.split()[0]
            scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1))
            spill = int(re.search(r"VGPRs Spill: (\d+)", blk).group(1))
            seen += 1
            assert scratch == 0 and spill == 0, f"{src}: {name} uses scratch ({scratch} B/lane, {spill} spilled VGPRs)"
    assert seen >= 10  # every template instantiation was looked at
