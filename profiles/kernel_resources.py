#!/usr/bin/env python3
"""Registers / spills / LDS of every kernel in a built object or library (no GPU needed):
    python profiles/kernel_resources.py collaborative-filtering_amd/csrc/row_solve.o [name filter]
Pulls the gfx950 code object out of the clang offload bundle and reads its metadata notes with llvm-readelf."""
import re
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    blob = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = blob.find(magic)
    while pos >= 0:
        n = int.from_bytes(blob[pos + 24:pos + 32], "little")
        p = pos + 32
        for _ in range(n):
            off, size, tl = (int.from_bytes(blob[p + 8 * j:p + 8 * j + 8], "little") for j in range(3))
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx" in triple and size:
                yield triple, blob[pos + off:pos + off + size]
        pos = blob.find(magic, pos + 1)


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for triple, elf in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda key: (re.search(r"\." + key + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            if flt in name:
                agpr = blk.split()[0]
                print(f"vgpr {g('vgpr_count'):>4} agpr {agpr:>4} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>4} "
                      f"scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}  "
                      f"{name[:110]}")


if __name__ == "__main__":
    main()
