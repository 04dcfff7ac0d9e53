"""Compile-time guard for the hot kernels (no GPU needed: hipcc cross-compiles): none of the propagate kernels may
spill registers to scratch memory.  (A spill does not change results - the GPU suite stays green - but it made the
bench kernel 12x slower once, when two lambdas of k_zpropagate3 stopped being inlined.)"""
import os
import re
import subprocess
import tempfile

from imcoalhmm_amd import build

HOT = ("k_zpropagate4", "k_z4_level", "k_z4_level2", "k_zpropagate3", "k_zpropagate2", "k_zpropagate", "k_propagate", "k_big_propagate", "k_big_vector", "k_chain")


def test_hot_kernels_do_not_spill():
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [build.hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
               "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "k.o"), build.SRC]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
    name, seen, bad = None, 0, []
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and any(("%d%s" % (len(h), h)) in name for h in HOT):     # mangled: <length><identifier>
            seen += 1
            # the 16-wavefront vector token kernel (24 < N <= 64, few chunks x many proposals) has had 36-152 bytes
            # of prologue spill in four of its shapes since round 1 (128-register budget); everything else: none
            limit = 192 if "12k_zpropagateI" in name else 0
            # 21-24 states: P, Q and one operand set are 3 x 72 registers of the 256 - the streamed form spills 27-29 VGPRs,
            # the hybrid one (tests only) up to 72, and the launch is still 1.9-2.3x faster than the 32-token LDS table it
            # replaced (profiles/r03_h_states24_levels.txt)
            if "13k_zpropagate4ILi6E" in name:
                limit = 320
            if int(m.group(1)) > limit:
                bad.append((name, int(m.group(1))))
    assert seen >= 40, seen            # every shape of every family was looked at
    assert not bad, bad
