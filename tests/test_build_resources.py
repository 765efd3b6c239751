"""Compile-time resource guard (CPU only, hipcc cross-compiles): no kernel may use scratch memory or spill registers.
A 16-byte struct copy through the wrong pointer type once put a tile fragment in scratch and cost 35 % throughput."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

from fgt_amd import build as B


def _usage(src):
    # the report build() recorded when it compiled this source (same flags); compiled here only when the object is missing or stale
    txt = B.usage_report(src)
    if txt is None:
        cmd = [B._hipcc()] + B.FLAGS + ["-c", os.path.join(B.CSRC, src), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        txt = r.stdout + r.stderr
    names = re.findall(r"Function Name: (\S+)", txt)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", txt)]
    # (round 2 exempted the 8-phase 256x256 tiles here: hipcc declined to unroll the epilogue's row-block loop with 8 accumulator tiles per
    #  wavefront and indexed the accumulators in scratch; conv_tile.h's static_for made the index a compile-time constant: no exemptions)
    return scratch, [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", txt)]


def test_no_kernel_uses_scratch_or_spills():
    srcs = [s for s in B.SOURCES if s != "runtime.hip"]
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = dict(zip(srcs, ex.map(_usage, srcs)))
    for src, (scratch, spills) in res.items():
        assert scratch, f"{src}: no kernels reported"
        allowed = 40 if src == "conv_igemm.hip" else 0     # the register-staged 8-wave 128x128 bf16x3 tile is capped at 128 VGPRs (<= 10 dwords spill; the hot path runs on conv_split.hip)
        assert max(scratch) <= allowed and sum(1 for s in scratch if s) <= 1, f"{src}: scratch bytes/lane {scratch}"
        assert sum(spills) <= 10, f"{src}: VGPR spills {spills}"
