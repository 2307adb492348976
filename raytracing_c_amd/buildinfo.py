"""Identity of the path kernel's build: a hash over the sources the tile-stream kernel is compiled from and the compiler flag
line.  tools/summarize_profile.py stamps profiles/*_traffic.json with it; bench.py compares the stamp with the tree before it
replays PMC counters of a committed profile (a process cannot read the counters of its own kernels), and refuses them when a
single character of the kernel changed since the profile was taken."""
import hashlib
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
KERNEL_SOURCES = ("csrc/rt_kernels.hip", "csrc/rt_dev.hip.h", "csrc/rt_device.h", "../include/rt_math.h")


def hipflags():
    txt = open(os.path.join(_HERE, "csrc", "Makefile")).read()
    m = re.search(r"^HIPFLAGS\s*:=\s*(.*)$", txt, re.M)
    flags = m.group(1).strip() if m else ""
    r = re.search(r"^RAFLAGS\s*:=\s*(.*)$", txt, re.M)           # (the register-allocation flags HIPFLAGS includes by name)
    return flags.replace("$(RAFLAGS)", r.group(1).strip() if r else "")


def kernel_source_hash():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(_HERE, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    h.update(hipflags().encode())
    return h.hexdigest()[:16]
