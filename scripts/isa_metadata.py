"""Per-kernel code-object metadata of a built libpbrhip*.so: registers, LDS, scratch, code size.

  python scripts/isa_metadata.py [library.so] [--json out.json] [--disasm dir]

Takes the gfx950 code object out of the library's fat binary (llvm-objcopy + clang-offload-bundler), reads the
AMDGPU metadata notes (llvm-readelf --notes) and the symbol table (kernel code sizes), and prints one row per kernel.
Static LDS is 0 for the path-tracing kernels: their LDS (the staged tree top, the drain ring) is dynamic — sized per
launch by the host (pbr_hip.hip, makePlan) — which is why rocprofv3's kernel trace shows LDS_Block_Size 0 for them
unless it adds the dynamic part, and why the trace's VGPR_Count is the allocation granule-rounded accum_offset half
on gfx950's unified register file (see profiles/r03/isa/README.md).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(lib, workdir):
    """Every gfx950 code object of the library: since round 5 it is linked from one translation unit per plan and build
    flavour (build.py), each with its own offload bundle in .hip_fatbin."""
    fat = os.path.join(workdir, "fat.bin")
    # (with an explicit output file: without one llvm-objcopy rewrites its INPUT in place — same contents, new layout and mtime)
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(workdir, "copy.so")])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for k, start in enumerate(starts):
        part = os.path.join(workdir, "bundle%d.bin" % k)
        with open(part, "wb") as f:
            f.write(blob[start:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        dev = os.path.join(workdir, "dev%d.co" % k)
        # the product is built for gfx950:xnack- since round 4 (build.py, HIP_FLAGS); older and lab libraries for plain gfx950
        for target in ("hipv4-amdgcn-amd-amdhsa--gfx950:xnack-", "hipv4-amdgcn-amd-amdhsa--gfx950"):
            done = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                                   "--targets=" + target, "--output=" + dev], capture_output=True, text=True)
            if done.returncode == 0 and os.path.exists(dev) and os.path.getsize(dev) > 0:
                out.append(dev)
                break
    if not out:
        raise RuntimeError("no gfx950 code object in %s" % lib)
    return out


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def kernels(dev):
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", dev], capture_output=True, text=True).stdout
    recs, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+(- )?\.([a-z_]+):\s+(.*)$", line)
        if not m:
            continue
        dash, key, val = m.groups()
        if key == "agpr_count" and dash:
            cur = {}
            recs.append(cur)
        if cur is not None and key in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                                       "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size", "name", "kernarg_segment_size",
                                       "uses_dynamic_stack", "wavefront_size"):
            cur[key] = val.strip().strip("'") if key == "name" else (val.strip() if key == "uses_dynamic_stack" else int(val))
    syms = subprocess.run([LLVM + "/llvm-readelf", "--symbols", "--wide", dev], capture_output=True, text=True).stdout
    sizes = {}
    for line in syms.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            sizes[f[7]] = int(f[2])
    names = demangle([r["name"] for r in recs])
    for r in recs:
        r["code_bytes"] = sizes.get(r["name"], 0)
        r["kernel"] = names[r["name"]]
        # unified register file of a gfx950 SIMD: 512 per lane; allocation granule 8
        regs = ((r["vgpr_count"] + 7) // 8) * 8 + ((r["agpr_count"] + 7) // 8) * 8
        r["waves_per_simd_by_registers"] = min(8, 512 // max(regs, 8))
    return recs


def main():
    argv, args, skip = sys.argv[1:], [], False
    for a in argv:
        if skip:
            skip = False
        elif a in ("--json", "--disasm"):
            skip = True
        else:
            args.append(a)
    lib = args[0] if args else os.path.join(ROOT, "physically-based-rendering_amd", "csrc", "libpbrhip.so")
    with tempfile.TemporaryDirectory() as tmp:
        devs = code_objects(lib, tmp)
        recs = [r for dev in devs for r in kernels(dev)]
        if "--disasm" in sys.argv:
            out = sys.argv[sys.argv.index("--disasm") + 1]
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "gfx950.s"), "w") as f:
                for dev in devs:
                    subprocess.check_call([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", dev], stdout=f)
    recs.sort(key=lambda r: r["kernel"])
    print("%-100s %5s %5s %5s %8s %8s %6s %9s %5s" % ("kernel", "VGPR", "AGPR", "SGPR", "LDS(st.)", "scratch", "spills", "code B", "w/SIMD"))
    for r in recs:
        print("%-100s %5d %5d %5d %8d %8d %6d %9d %5d" % (
            r["kernel"][:100], r["vgpr_count"], r["agpr_count"], r["sgpr_count"], r["group_segment_fixed_size"],
            r["private_segment_fixed_size"], r["vgpr_spill_count"], r["code_bytes"], r["waves_per_simd_by_registers"]))
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(recs, f, indent=1)


if __name__ == "__main__":
    main()
