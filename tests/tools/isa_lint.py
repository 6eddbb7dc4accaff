"""TEST-ONLY lint of gfx950 code objects for one miscompile pattern of the ROCm 7.2 compiler, found with rocgdb on
k_adjoint_tsit5: register-spill copies (v_accvgpr_write/read, scratch_store/load) placed at the top of
a control-flow join block BEFORE the `s_or_b64 exec, exec, s[..]` that re-enables the lanes which skipped the region.  The
copies then run under the region's narrower exec mask — with an empty one when the join is reached through the region's
`s_cbranch_execz` — so the skipped lanes' values are never saved (or never restored).

    python tests/tools/isa_lint.py <code object | shared library with a clang offload bundle> ...

prints one line per flagged site and exits 1 if there is any."""
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
SPILL = ("v_accvgpr_write", "v_accvgpr_read", "scratch_store", "scratch_load", "buffer_store", "buffer_load")


def extract_gfx950(path):
    """Every gfx950 code object of `path`: the file itself when it is a plain code object, else one temporary file per
    clang offload bundle (a library linked from several translation units carries one bundle per unit)."""
    d = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = d.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", d, i + 24)[0]
        off = i + 32
        pos = i + 24
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", d, off)
            off += 24
            triple = d[off:off + tl]
            off += tl
            if b"gfx950" in triple and sz > 0:
                f = tempfile.NamedTemporaryFile(suffix=".hsaco", delete=False)
                f.write(d[i + o:i + o + sz])
                f.close()
                out.append(f.name)
            pos = max(pos, i + o + sz)
    if not out:
        if d[:4] == b"\x7fELF" and b"__CLANG_OFFLOAD_BUNDLE__" not in d:
            return [path]
        raise RuntimeError("no gfx950 code object in " + path)
    return out


def disassemble(path):
    return "\n".join(subprocess.check_output([OBJDUMP, "-d", "--no-show-raw-insn", co]).decode() for co in extract_gfx950(path))


def lint(path):
    txt = disassemble(path)
    kernels, cur = [], None
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = (m.group(1), [])
            kernels.append(cur)
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-F]+):", line)
        if m and cur is not None:
            op, addr = m.group(1), int(m.group(2), 16)
            t = re.search(r"<.*\+0x([0-9a-f]+)>", line) if op.startswith(("s_cbranch", "s_branch")) else None
            cur[1].append((addr, op, int(t.group(1), 16) if t else None))
    findings = []
    STOP = ("s_cbranch", "s_branch", "s_endpgm", "s_or_b64 exec", "s_and_saveexec", "s_andn2_saveexec", "s_or_saveexec", "s_mov_b64 exec", "s_andn2_b64 exec", "s_xor_b64 exec")
    for name, insns in kernels:
        if not insns:
            continue
        base = insns[0][0]
        execz_targets = {base + t for a, op, t in insns if t is not None and op.startswith("s_cbranch_execz")}
        targets = {base + t for a, op, t in insns if t is not None}
        for k, (a, op, _) in enumerate(insns):
            if not op.startswith("s_or_b64 exec, exec"):
                continue
            j, spills = k - 1, []
            while j >= 0:
                aj, opj, _t = insns[j]
                if opj.startswith(STOP):
                    break
                if opj.startswith(SPILL):
                    spills.append(" ".join(opj.split()[:2]).rstrip(","))
                if aj in targets:
                    if spills and aj in execz_targets:
                        findings.append((name, hex(a), len(spills), spills[-1]))
                    break
                j -= 1
    return findings


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for name, addr, n, first in lint(p):
            print(f"{p}: {name[:90]} @ {addr}: {n} spill copies ahead of the exec restore (first: {first})")
            bad += 1
    sys.exit(1 if bad else 0)
