#!/usr/bin/env python3
"""What the SILICON is told about every kernel of librolo_hip.so: register counts, spills, LDS and scratch from the code objects' own metadata (the amdhsa.kernels notes),
not from rocprofv3's VGPR_Count column — that column is the allocation granule count of the ARCH VGPR file only (58 / 202 / 192 / 110 / 146 registers showed up as
32 / 104 / 96 / 56 / 76 in round 5's sq_counters.csv: round 5's verdict, item 6).

    python profiles/tools/codeobj_notes.py [rolo_amd/librolo_hip.so] > profiles/rNN/codeobj_notes.csv

Finds the clang offload bundles inside the shared object (magic __CLANG_OFFLOAD_BUNDLE__: entry table of offset / size / triple), writes each gfx950 code object to a
temporary file and reads `llvm-readelf --notes` (the metadata is msgpack, printed as YAML). waves_per_simd = what the register file allows: 512 registers per SIMD lane
on gfx950 (unified VGPR + AGPR file), allocated in blocks of 8, at most 8 wavefronts."""
import csv
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    blob = open(path, "rb").read()
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size > 0:
                yield triple, blob[pos + off:pos + off + size]
        pos = q


def demangle(names):
    """c++filt in one go; falls back to the mangled names"""
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return dict(zip(names, out)) if len(out) == len(names) else {n: n for n in names}
    except Exception:
        return {n: n for n in names}


def short(dem):
    """`knn_walk_kernel<20, false, false, true>` from `void rolo::(anonymous namespace)::knn_walk_kernel<20, false, false, true>(rolo::KnnPair, int, int, int)` — the same
    form summarize_sq.py makes of rocprofv3's Kernel_Name"""
    m = re.search(r"([A-Za-z_0-9]+)(<[^()]*>)?\(", dem)
    return (m.group(1) + (m.group(2) or "")) if m else dem


def kernels(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes); f.flush()
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    for blk in re.split(r"\n\s*- \.agpr_count:", "\n" + txt)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k, d="0": (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, d])[1]
        name = g("name", "?")
        yield dict(kernel=name, mangled=name, vgpr=int(g("vgpr_count")), agpr=int(g("agpr_count")), sgpr=int(g("sgpr_count")), vgpr_spill=int(g("vgpr_spill_count")),
                   sgpr_spill=int(g("sgpr_spill_count")), lds_bytes=int(g("group_segment_fixed_size")), scratch_bytes_per_lane=int(g("private_segment_fixed_size")),
                   max_workgroup=int(g("max_flat_workgroup_size")))


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "rolo_amd", "librolo_hip.so")
    rows = {}
    for triple, elf in code_objects(path):
        for k in kernels(elf):
            rows[k["mangled"]] = k
    dm = demangle(list(rows))
    for m_, k in rows.items():
        k["kernel"] = short(dm[m_])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill", "sgpr_spill", "lds_bytes", "scratch_bytes_per_lane", "max_workgroup", "waves_per_simd_by_registers", "mangled"])
    for k in sorted(rows.values(), key=lambda r: r["kernel"]):
        regs = max(k["vgpr"] + k["agpr"], 1)
        blocks = -(-regs // 8) * 8
        w.writerow([k["kernel"], k["vgpr"], k["agpr"], k["sgpr"], k["vgpr_spill"], k["sgpr_spill"], k["lds_bytes"], k["scratch_bytes_per_lane"], k["max_workgroup"], min(8, 512 // blocks), k["mangled"]])


if __name__ == "__main__":
    main()
