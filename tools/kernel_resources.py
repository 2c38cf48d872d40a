#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in libviprs_hip.so, read from the code object's metadata (no GPU needed):
    python tools/kernel_resources.py [lib.so] [--min-scratch N]
The gfx950 code object is taken out of the library's .hip_fatbin with clang-offload-bundler; `llvm-readelf --notes` prints
the AMDGPU metadata (one record per kernel: .vgpr_count, .agpr_count, .sgpr_count, .private_segment_fixed_size = scratch
bytes per lane, .group_segment_fixed_size = static LDS)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(lib):
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        # one bundle per translation unit, concatenated in the section
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
        notes = ""
        for i in range(len(starts) - 1):
            part, co = os.path.join(td, f"b{i}.bin"), os.path.join(td, f"b{i}.co")
            open(part, "wb").write(blob[starts[i]:starts[i + 1]])
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--output={co}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True, stderr=subprocess.DEVNULL)
            notes += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = []
    for rec in notes.split("- .agpr_count:")[1:]:
        rec = ".agpr_count:" + rec
        get = lambda k: (re.search(rf"\.{k}:\s*(\S+)", rec) or [None, "0"])[1]
        name = get("name")
        dem = name
        out.append(dict(name=dem, vgpr=int(get("vgpr_count")), agpr=int(get("agpr_count")), sgpr=int(get("sgpr_count")),
                        scratch=int(get("private_segment_fixed_size")), lds=int(get("group_segment_fixed_size")),
                        spill_v=int(get("vgpr_spill_count")), spill_s=int(get("sgpr_spill_count"))))
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    min_scratch = 0
    if "--min-scratch" in argv:
        i = argv.index("--min-scratch")
        min_scratch = int(argv[i + 1])
        del argv[i:i + 2]
    lib = argv[0] if argv else os.path.join(ROOT, "viprs_amd", "lib", "libviprs_hip.so")
    ks = kernels(lib)
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in ks), capture_output=True, text=True).stdout.splitlines()
    for k, n in zip(ks, names):
        k["name"] = n
    ks = sorted(ks, key=lambda k: (-k["scratch"], k["name"]))
    print(f"{len(ks)} kernels in {lib}; scratch > 0: {sum(k['scratch'] > 0 for k in ks)}, max {max(k['scratch'] for k in ks)} B per lane")
    for k in ks:
        if k["scratch"] >= min_scratch:
            print(f"scratch {k['scratch']:4d} B  vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d}  spills v{k['spill_v']} s{k['spill_s']}  lds {k['lds']:6d}  {k['name'][:150]}")
