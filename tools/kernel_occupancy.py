#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in the built objects (stringwars_amd/csrc/*.o), and the waves per SIMD they leave.

    tools/kernel_occupancy.py            # print the table
    tools/kernel_occupancy.py --write    # ... and write profiles/r6/kernel_resources.json (tests/test_bench_tools.py compares the
                                         # build against it: a kernel that silently drops an occupancy class fails the CPU suite)

Why: round 6 lost 1.6 x on the ACGT-100 cross-product to ONE register -- `misfit |= 1` in place of `misfit = 1` took
k_align_cross_wide<128> from 256 to 257 VGPRs, two waves per SIMD to one -- and nothing but a benchmark table noticed.
No GPU needed: the device code objects are unbundled from the objects' .hip_fatbin sections."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stringwars_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
TABLE = os.path.join(ROOT, "profiles", "r6", "kernel_resources.json")
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count")


def waves_per_simd(vgprs, agprs=0):
    """gfx950: 512 registers per SIMD lane shared by a wave's VGPRs and AGPRs, allocated in blocks of eight, at most eight waves."""
    blocks = (max(vgprs + agprs, 1) + 7) // 8 * 8
    return min(8, 512 // blocks)


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat"), os.path.join(tmp, "co")
        done = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True, text=True)
        if done.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return {}
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    out, record = {}, {}
    for line in notes.splitlines():
        m = re.search(r"\.(name|" + "|".join(FIELDS) + r"):\s+(\S+)", line)
        if not m:
            continue
        key, value = m.groups()
        if key == "name":
            if record.get("name") and "vgpr_count" in record:
                out[record.pop("name")] = record
            record = {"name": value} if value.startswith("_Z") and not value.endswith(".kd") else {}
        elif "name" in record:
            record[key] = int(value)
    if record.get("name") and "vgpr_count" in record:
        out[record.pop("name")] = record
    for entry in out.values():
        entry["waves_per_simd"] = waves_per_simd(entry.get("vgpr_count", 0), entry.get("agpr_count", 0))
    return out


def demangled(names):
    try:
        done = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    except OSError:
        return {n: n for n in names}
    return dict(zip(names, done.stdout.splitlines())) if done.returncode == 0 and names else {n: n for n in names}


def build_table():
    table = {}
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(".o"):
            found = kernels_of(os.path.join(CSRC, name))
            pretty = demangled(list(found))
            for mangled, entry in found.items():
                table[re.sub(r"\(.*", "", pretty[mangled]).replace("void ", "")] = dict(entry, object=name)
    return table


def main():
    table = build_table()
    for kernel, e in sorted(table.items(), key=lambda kv: (kv[1]["object"], kv[0])):
        print(f"{e['object']:<14} {kernel[:96]:<96} vgpr {e.get('vgpr_count', 0):3d} agpr {e.get('agpr_count', 0):3d} lds {e.get('group_segment_fixed_size', 0):6d} "
              f"scratch {e.get('private_segment_fixed_size', 0):4d} spill {e.get('vgpr_spill_count', 0):3d} waves/SIMD {e['waves_per_simd']}")
    if "--write" in sys.argv:
        os.makedirs(os.path.dirname(TABLE), exist_ok=True)
        json.dump({"about": "tools/kernel_occupancy.py --write: resources of every kernel of the built objects (gfx950); waves_per_simd from the registers alone",
                   "kernels": table}, open(TABLE, "w"), indent=1, sort_keys=True)
        print("wrote", os.path.relpath(TABLE, ROOT), len(table), "kernels")


if __name__ == "__main__":
    main()
