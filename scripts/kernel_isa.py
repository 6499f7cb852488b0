"""Compile ONE instantiation of a step kernel for gfx950 (no GPU needed, a few seconds) and print what the compiler made of it:
registers, spills, scratch, and the memory instructions / waits / barriers of its first lines - the view DESIGN.md 3.1c's
last bullets were read off.

python scripts/kernel_isa.py 'step_kernel_spec_multi<SpecBalance4, 0, ENV_BALANCE, DevEnv, true>' [--head 120] [--asm out.s]
python scripts/kernel_isa.py 'step_kernel_spec<SpecBalance4, 0>'

The unit is csrc/vmas_hip.hip up to the end of its kernel section plus one explicit instantiation; flags as csrc/build.sh."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "vectorizedmultiagentsimulator_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "--cuda-device-only"]
SIGS = {  # kernel name -> parameter list of the explicit instantiation
    "step_kernel_spec_multi": "(DevWorld, float*, float*, long, int, int, long, const {env}, LazyArgs)",
    "step_kernel_spec": "(DevWorld, float*, float*, long, int, LazyArgs)",
    "step_kernel": "(DevWorld, float*, float*, long, int, DevStepArgs, const {env})",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", help="e.g. 'step_kernel_spec_multi<SpecBalance4, 0, ENV_BALANCE, DevEnv, true>'")
    ap.add_argument("--head", type=int, default=80, help="memory instructions / waits to list")
    ap.add_argument("--asm", help="keep the assembly here")
    a = ap.parse_args()
    name = a.kernel.split("<")[0].strip()
    if name not in SIGS:
        sys.exit(f"kernel_isa: one of {sorted(SIGS)}")
    env = "DevEnv" if "DevEnv" in a.kernel else "NoEnv"
    src = open(os.path.join(CSRC, "vmas_hip.hip")).read()
    cut = src.index('#include "vmas_compact.h"')  # everything the step kernels need is in front of this line
    unit = src[:cut] + f"\ntemplate __global__ void {a.kernel}{SIGS[name].format(env=env)};\n"
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(CSRC, "_kernel_isa_unit.hip")  # (beside the headers it includes)
        open(path, "w").write(unit)
        try:
            out = a.asm or os.path.join(tmp, "k.s")
            r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-S", path, "-o", out],
                               capture_output=True, text=True)
        finally:
            os.remove(path)
        if r.returncode != 0:
            sys.exit(r.stderr[-3000:])
        for line in r.stderr.splitlines():
            m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[", line)
            if m:
                print(f"{m.group(1):30s} {m.group(2)}")
        asm = open(out).read().splitlines()
    start = next(i for i, l in enumerate(asm) if re.match(r"^_Z\w+:", l))
    body = asm[start:]
    end = next((i for i, l in enumerate(body) if "s_endpgm" in l), len(body))
    body = body[:end]
    count = lambda pat: sum(1 for l in body if re.search(pat, l))
    print(f"{'instructions (lines)':30s} {len(body)}")
    print(f"{'scratch loads / stores':30s} {count(r'scratch_load')} / {count(r'scratch_store')}")
    print(f"{'flat loads / stores':30s} {count(r'flat_load')} / {count(r'flat_store')}   (count on lgkmcnt as well as vmcnt)")
    print(f"{'v_readlane / v_writelane':30s} {count(r'v_readlane')} / {count(r'v_writelane')}   (scalar spills live in vector lanes)")
    print(f"-- the first {a.head} memory instructions, waits and barriers (line of the kernel: instruction)")
    n = 0
    for i, l in enumerate(body):
        if re.search(r"global_load|global_store|flat_|scratch_|s_load_|s_waitcnt|s_barrier|buffer_|ASMSTART", l):
            print(f"{i:6d}: {l.strip().split(';')[0][:100]}")
            n += 1
            if n >= a.head:
                break


if __name__ == "__main__":
    main()
