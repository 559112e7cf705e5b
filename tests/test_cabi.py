"""The drop-in boundary without a GPU: the shared library loads, exports every symbol that
include/pbr_hip.h declares, the ctypes mirrors have the reference's struct sizes, and — with no
device — pbr_create fails loudly instead of falling back to anything."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pbr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pbr_[a-z_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_symbols()
    for must in ("pbr_create", "pbr_destroy", "pbr_upload_scene", "pbr_configure", "pbr_render_frame",
                 "pbr_render", "pbr_accumulate", "pbr_read_output", "pbr_read_debug", "pbr_last_error",
                 "pbr_last_kernel_ms", "pbr_export_tiles", "pbr_import_tiles", "pbr_get_counters"):
        assert must in names


def test_library_exports_every_declared_symbol(pbr):
    for name in declared_symbols():
        assert hasattr(pbr.hip, name), "libpbrhip.so does not export %s" % name


def test_abi_version_is_checked(pbr):
    """pbr_config grew in round 5 with nothing a caller could check: the header carries PBR_ABI_VERSION, the
    library answers pbr_abi_version(), the harness refuses a mismatch at import."""
    header = open(os.path.join(ROOT, "include", "pbr_hip.h")).read()
    declared = int(re.search(r"#define PBR_ABI_VERSION (\d+)", header).group(1))
    assert pbr.hip.pbr_abi_version() == declared == pbr.ABI_VERSION


def test_wire_struct_sizes_match_the_reference(pbr):
    # camera_cl 80 B, bvhNode_cl 32 B, light_cl 48 B, materials 48 / 64 B (source/PathTracer.h:25-73)
    assert ctypes.sizeof(pbr.Camera) == 80
    assert ctypes.sizeof(pbr.Float4) == 16
    assert ctypes.sizeof(pbr.Counters) == 32
    assert ctypes.sizeof(pbr.Config) == 68     # 60 + traversal + arith (round 5)


def test_no_device_means_failure_not_fallback(pbr):
    """Only meaningful where there is no GPU (the build container); on a GPU box creation works."""
    ctx = ctypes.c_void_p()
    status = pbr.hip.pbr_create(0, ctypes.byref(ctx))
    try:
        if status == 0:
            pytest.skip("a HIP device is present")
        assert status == -2                                              # PBR_EDEVICE
        msg = pbr.hip.pbr_last_error(ctx).decode()
        assert "no HIP device" in msg and "no CPU path" in msg
        # nothing else works on the dead context either
        assert pbr.hip.pbr_reset_accum(ctx) != 0
    finally:
        if ctx:
            pbr.hip.pbr_destroy(ctx)
    with pytest.raises(pbr.PbrError):
        pbr.Device(0)


def test_device_code_is_built_for_gfx950_without_xnack(pbr):
    """The hand-scheduled node phase lets a record load overwrite its own address register (csrc/pt_kernel.hpp,
    PT_NODE_PHASE_HEAD), which is only legal where a faulted load is never replayed: the code objects must say
    gfx950:xnack-, so that the loader refuses them in an XNACK-enabled process instead of running them there."""
    from importlib import import_module
    build = import_module(pbr.__name__ + ".build")
    libs = [build.HIP_LIB] + ([build.HIP_GUARD_LIB] if os.path.exists(build.HIP_GUARD_LIB) else [])
    for lib in libs:
        blob = open(lib, "rb").read()
        targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+(?::[a-z+\-]+)*)", blob))
        assert targets == {b"gfx950:xnack-"}, (lib, targets)


def test_product_does_not_touch_the_oracle():
    """The oracle is test infrastructure: nothing in the package or the headers may name it."""
    pkg = os.path.join(ROOT, "physically-based-rendering_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hpp", ".hip")):
                text = open(os.path.join(base, f), errors="ignore").read().lower()
                assert "pt_oracle" not in text and "liboracle" not in text and "import oracle" not in text, f


def test_scene_validation_without_a_device(pbr):
    """pbr_validate_scene = the checks of pbr_upload_scene, context-free.  A miss link that points backward (or at the
    node itself) would make the stackless walk circle forever on the device: rejected before anything is uploaded."""
    import numpy as np
    pbr.cfg_reset()
    sc = pbr.HostScene.generate("sponza", 3, 3000)
    assert pbr.validate_scene(sc.desc) == ""
    arr = sc.arrays()
    inner = np.nonzero((arr["bvh"][:, 3] == -1.0) & (arr["bvh"][:, 7] > 0))[0]
    assert len(inner) > 10

    def with_bvh(nodes):
        d = pbr.SceneDesc.from_buffer_copy(sc.desc)
        d.bvh = nodes.ctypes.data
        return d

    i = int(inner[5])
    for bad_link in (i, i - 1, 1):                                       # itself, the node before, node 1
        nodes = arr["bvh"].copy()
        nodes[i, 7] = bad_link
        assert "must point forward" in pbr.validate_scene(with_bvh(nodes)), bad_link
    for fine in (-1.0, 0.0, float(i + 1)):                               # end of the walk, or forward
        nodes = arr["bvh"].copy()
        nodes[i, 7] = fine
        assert pbr.validate_scene(with_bvh(nodes)) == ""
    nodes = arr["bvh"].copy()
    nodes[i, 7] = 2.5
    assert "not an index" in pbr.validate_scene(with_bvh(nodes))
    nodes = arr["bvh"].copy()
    nodes[-1, 3], nodes[-1, 7] = -1.0, -1.0
    assert "last node is a container" in pbr.validate_scene(with_bvh(nodes))
    faces = arr["facesV"].copy()
    faces[7, 3] = 0xFFFFFFFF
    d = pbr.SceneDesc.from_buffer_copy(sc.desc)
    d.facesV = faces.ctypes.data
    assert "material index" in pbr.validate_scene(d)
    faces = arr["facesV"].copy()
    faces[7, 1] = sc.desc.num_vertices
    d.facesV = faces.ctypes.data
    assert "vertex index" in pbr.validate_scene(d)
    d = pbr.SceneDesc.from_buffer_copy(sc.desc)
    d.num_nodes = 1
    assert "root is never tested" in pbr.validate_scene(d)


def test_entry_points_refuse_a_missing_context(pbr):
    """Every entry point that takes a context returns an error for NULL (no crash, no fallback), the round-2 additions
    included; pbr_bvh_node_capacity is context-free: 2 n - 1 records at most, 2 at least (a root above a single leaf)."""
    hip = pbr.hip
    null = ctypes.c_void_p()
    cam, par = pbr.Camera(), pbr.DenoiseParams()
    buf = (ctypes.c_float * 16)()
    count = ctypes.c_uint32()
    assert hip.pbr_denoise(null, 0.01, ctypes.byref(cam), ctypes.byref(par), buf, None) != 0
    assert hip.pbr_build_bvh(null, None, 0, None, None, 0, None, ctypes.byref(count), None, None) != 0
    assert hip.pbr_reset_accum(null) != 0 and hip.pbr_read_output(null, buf) != 0 and hip.pbr_read_full(null, buf) != 0
    assert hip.pbr_render_frame(null, 0.5, 0.5, 0.01, ctypes.byref(cam)) != 0
    assert hip.pbr_diag_pin_plan(null, 0) != 0
    assert hip.pbr_diag_set_knob(null, b"lds_slots", 0) != 0
    assert [hip.pbr_bvh_node_capacity(n) for n in (0, 1, 2, 3, 1000)] == [2, 2, 3, 5, 1999]
    p = pbr.DenoiseParams()
    assert (p.passes, round(p.sigma_color, 3), round(p.sigma_normal, 3), round(p.sigma_world, 3), round(p.sigma_albedo, 3)) == (5, 1.2, 0.25, 3.0, 0.1)
    assert ctypes.sizeof(pbr.DenoiseParams) == 20


def test_concurrent_builders_never_publish_a_half_written_library(tmp_path):
    """The ranks of a multi-process job all import the package and may all find a library stale: builds are serialised by
    a file lock, written aside and moved into place, so every process loads a complete library."""
    import subprocess
    import sys
    code = (
        "import sys, ctypes; sys.path.insert(0, %r)\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('pbr_build', %r)\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "lib = b.build_host(force=True)\n"
        "ctypes.CDLL(lib, mode=ctypes.RTLD_GLOBAL).pbrh_cfg_reset()\n"
        "print('loaded')\n" % (ROOT, os.path.join(ROOT, "physically-based-rendering_amd", "build.py")))
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(3)]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0 and "loaded" in out, err[-2000:]
    leftovers = [f for f in os.listdir(os.path.join(ROOT, "physically-based-rendering_amd", "host")) if f.endswith(".tmp")]
    assert not leftovers


def test_the_product_library_reads_no_environment_variable(pbr):
    """VERDICT r02 hygiene item: a stray PBR_* variable in a viewer's environment must not change how the library
    renders.  No source of libpbrhip.so calls getenv (experiment knobs are per-context values behind pbr_diag_set_knob;
    the Python harness maps environment variables onto them).  (The library does import the symbol: rocPRIM's headers,
    pulled in by the device BVH build's radix sort, read their own tuning variables — none of them is this path's.)"""
    csrc = os.path.join(ROOT, "physically-based-rendering_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_reference_flavour_kernels_are_round_4s_register_for_register():
    """Round 5 added two opt-in modes as BUILD FLAVOURS precisely so that the reference mode's kernels would not change
    (csrc/pt_flavour.hpp: as run-time branches the modes cost the 80-register state machine its scratch-free build).  The
    committed code-object metadata says they did not: every path-tracing kernel of flavour 0 (profiles/r05/isa) has the
    VGPRs, the scratch and the spilled registers of its round-4 namesake (profiles/r04/isa), and the ordered flavour's
    80-register state machine is scratch-free too."""
    import json
    r4 = {k["kernel"]: k for k in json.load(open(os.path.join(ROOT, "profiles", "r04", "isa", "kernels.json"))) if "pathTracing" in k["kernel"]}
    r5 = {k["kernel"]: k for k in json.load(open(os.path.join(ROOT, "profiles", "r05", "isa", "kernels.json"))) if "pathTracing" in k["kernel"]}
    seen = 0
    for name, old in r4.items():
        # ptk::pathTracingPhased<1, false, false, 6, 0>(ptk::DevParams) -> ptk_f0::pathTracingPhased<1, false, false, 6>(ptk_f0::DevParams)
        new_name = name.replace("ptk::", "ptk_f0::").replace(", 0>(", ">(")
        assert new_name in r5, name
        new = r5[new_name]
        for field in ("vgpr_count", "agpr_count", "private_segment_fixed_size", "vgpr_spill_count"):
            assert new[field] == old[field], (name, field, old[field], new[field])
        seen += 1
    assert seen == 48                                       # 8 groups x 6 variants
    flavours = {f: sum(1 for k in r5 if k.startswith("void ptk_f%d::" % f)) for f in range(4)}
    assert flavours == {0: 48, 1: 42, 2: 42, 3: 42}, flavours
    ordered_mid = r5["void ptk_f1::pathTracingPhased<1, false, false, 6>(ptk_f1::DevParams)"]
    assert ordered_mid["vgpr_count"] <= 80 and ordered_mid["private_segment_fixed_size"] == 0 and ordered_mid["vgpr_spill_count"] == 0


def test_round_6_left_the_kernels_in_use_register_for_register():
    """Round 6 changed one thing inside the path-tracing kernels of the existing flavours: the queue deals its tiles through a
    table (csrc/pt_kernel.hpp, nextSlot).  In kernels sized to the register that is not free by default — the first two
    formulations put 32 B of scratch into the 80-register kernels — so the committed code-object metadata is held to it: the
    kernels of the plans the tuner keeps on scenes without lights (refill-lean / -mid, phased-lean / -mid, phased-dual), in all
    four flavours of round 5, have round 5's VGPRs, scratch and spills (within a register where no scratch is involved); and
    the compact record's 6-waves state machine fits 80 registers without scratch like the eight streams' does."""
    import json
    r5 = {k["kernel"]: k for k in json.load(open(os.path.join(ROOT, "profiles", "r05", "isa", "kernels.json"))) if "pathTracing" in k["kernel"]}
    r6 = {k["kernel"]: k for k in json.load(open(os.path.join(ROOT, "profiles", "r06", "isa", "kernels.json"))) if "pathTracing" in k["kernel"]}
    seen = 0
    for f in range(4):
        for brdf in (0, 1):
            for shape in ("pathTracing<%d, false, false, 4, false>", "pathTracing<%d, false, false, 6, false>", "pathTracingPhased<%d, false, false, 4>",
                          "pathTracingPhased<%d, false, false, 6>", "pathTracingDual<%d, false, false>"):
                name = "void ptk_f%d::%s(ptk_f%d::DevParams)" % (f, shape % brdf, f)
                old, new = r5[name], r6[name]
                assert new["private_segment_fixed_size"] == old["private_segment_fixed_size"] == 0, (name, old["private_segment_fixed_size"], new["private_segment_fixed_size"])
                assert new["vgpr_spill_count"] == old["vgpr_spill_count"] == 0, name
                assert abs(new["vgpr_count"] - old["vgpr_count"]) <= 1, (name, old["vgpr_count"], new["vgpr_count"])
                seen += 1
    assert seen == 40
    flavours = {f: sum(1 for k in r6 if k.startswith("void ptk_f%d::" % f)) for f in range(8)}
    assert flavours == {0: 48, 1: 48, 2: 48, 3: 48, 4: 0, 5: 42, 6: 0, 7: 42}, flavours       # the Phong unit in every flavour; no two-paths kernel over compact records
    for f in (1, 5):
        mid = r6["void ptk_f%d::pathTracingPhased<1, false, false, 6>(ptk_f%d::DevParams)" % (f, f)]
        assert mid["vgpr_count"] <= 80 and mid["private_segment_fixed_size"] == 0 and mid["vgpr_spill_count"] == 0, f


def test_every_mode_is_built_into_the_product_library(pbr):
    """pbr_mode_built: the product library carries all six build flavours (reference / ray-ordered walk / ray-ordered walk over
    compact records x exact / native arithmetic); no device needed to ask."""
    for traversal in (0, 1, 2, 3):
        for arith in (0, 1):
            assert pbr.hip.pbr_mode_built(traversal, arith) == 1, (traversal, arith)
    assert pbr.hip.pbr_mode_built(4, 0) == -1 and pbr.hip.pbr_mode_built(0, 2) == -1
