#!/usr/bin/env python
"""Scan the device code of the built library for small loops that wait for every single global load before issuing the next
one (one or two loads, a wait for all outstanding loads, a backward branch): each iteration of such a loop costs a full memory
latency.  Prints kernel, loop length in instructions and loads per iteration.
   python tools/scan_serial_loads.py [path/to/libv2vae_hip.so]"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan(asm_path):
    cur, lines, seen = None, [], set()
    for ln in open(asm_path):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln.strip())
        if m:
            cur, lines = m.group(1), []
            continue
        mm = re.search(r"//\s*([0-9A-F]{12}):", ln)
        if cur is None or not mm:
            continue
        addr = int(mm.group(1), 16)
        ins = ln.strip().split("//")[0].strip()
        lines.append((addr, ins))
        b = re.match(r"s_cbranch_\w+\s+(\d+)", ins)
        if b and int(b.group(1)) >= 32768:
            target = addr + 4 + (int(b.group(1)) - 65536) * 4
            body = [i for a, i in lines if a >= target]
            loads = sum(i.startswith(("global_load", "flat_load", "buffer_load")) for i in body)
            waits = sum(i.startswith("s_waitcnt") and "vmcnt(0)" in i for i in body)
            if 0 < loads <= 2 and waits and len(body) < 40:
                key = (cur, len(body), loads)
                if key not in seen:
                    seen.add(key)
                    name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip() or cur
                    print(f"{name[:90]:90s} loop of {len(body):2d} instructions, {loads} load(s) per iteration")


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "volterra_vae_amd", "libv2vae_hip.so")
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], capture_output=True, cwd=tmp)
        for co in sorted(glob.glob(local + ".*amdgcn*")):
            asm = co + ".s"
            with open(asm, "w") as fh:
                subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], stdout=fh)
            scan(asm)
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
