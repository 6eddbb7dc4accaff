#!/usr/bin/env python
"""Static instruction mix of the gfx950 kernels in libhipadj.so (CPU-side tooling: llvm-objdump of the ROCm installation).

    python scripts/isa_count.py [substring of the demangled-ish kernel symbol ...]      default: the headline kernel k_interp<ModelLorenz, 8, 1>

For each matching kernel: instruction count, FP64 VALU ops (v_fma/mul/add_f64 ...), other VALU, SALU, memory ops, s_waitcnt, AGPR /
scratch spill traffic, and the largest VGPR index seen.  The headline kernel is FP64-issue bound at 10^4 trajectories (profiles/README.md:
4 cycles per wave-level FP64 instruction), so the FP64 count per unrolled block of 8 steps is the number to drive down."""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import isa_lint  # noqa: E402

pats = sys.argv[1:] or ["k_interpINS_11ModelLorenzELi8ELi1ELb1ELi1"]
txt = isa_lint.disassemble(os.path.join(ROOT, "scimlsensitivity.jl_amd", "libhipadj.so"))
cur, ker = None, {}
for line in txt.split("\n"):
    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
    if m:
        cur = m.group(1) if any(p in m.group(1) for p in pats) else None
        if cur:
            ker[cur] = []
        continue
    m = re.match(r"^\s+(\S.*?)\s*//", line)
    if m and cur:
        ker[cur].append(m.group(1))
for name, ins in ker.items():
    ops = collections.Counter(i.split()[0] for i in ins)
    f64 = sum(v for k, v in ops.items() if "_f64" in k and k.startswith("v_"))
    valu = sum(v for k, v in ops.items() if k.startswith("v_")) - f64
    salu = sum(v for k, v in ops.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_load", "s_nop")))
    mem = sum(v for k, v in ops.items() if k.startswith(("global_", "flat_", "buffer_", "ds_", "scratch_", "s_load")))
    spill = sum(v for k, v in ops.items() if k.startswith(("v_accvgpr", "scratch_")))
    vmax = max([int(x) for i in ins for x in re.findall(r"\bv\[?(\d+)", i)] or [0])
    print(f"{name[:100]}\n  instructions {len(ins)}  FP64 VALU {f64}  other VALU {valu}  SALU {salu}  memory {mem}  s_waitcnt {ops['s_waitcnt']}  spill ops {spill}  max VGPR v{vmax}")
    print("  FP64 mix:", {k: v for k, v in sorted(ops.items()) if "_f64" in k})
    print("  other VALU:", dict(collections.Counter({k: v for k, v in ops.items() if k.startswith("v_") and "_f64" not in k}).most_common(8)))
