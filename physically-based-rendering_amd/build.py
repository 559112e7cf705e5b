"""Builds the native parts in-tree.

  csrc/libpbrhip.so   HIP core + C ABI (hipcc, gfx950)             -- the product
  host/libpbrhost.so  C++ host side (BVH builder, loaders, driver) -- the product's caller side

Contraction is OFF everywhere (-ffp-contract=off): the path's arithmetic is defined
operation by operation, and the HIP kernels must reproduce it bit for bit.
"""
import contextlib
import fcntl
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, "include")
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")

HIP_LIB = os.path.join(CSRC, "libpbrhip.so")
HIP_GUARD_LIB = os.path.join(CSRC, "libpbrhip_guard.so")   # same source, -DPBR_GUARD: every device loop bounded
HOST_LIB = os.path.join(HOST, "libpbrhost.so")

MULTI_LIB = os.path.join(HOST, "libpbrmulti.so")          # N contexts in one process + RCCL (include/pbr_multi.h)
MULTI_SOURCES = ["multi_path_tracer.cpp"]
HOST_SOURCES = ["Cfg.cpp", "model_io.cpp", "bvh_builder.cpp", "scene_gen.cpp", "path_tracer.cpp", "cl_adaptor.cpp", "host_capi.cpp"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _digest(sources, flags=""):
    """Content hash of the sources (and the build flags): mtimes do not survive the copy to the GPU box."""
    import hashlib
    h = hashlib.sha256(flags.encode())
    for path in sorted(sources):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, sources, flags=""):
    """Missing, or built from other sources than the ones that are here now (the digest is kept next to the library)."""
    if not os.path.exists(target):
        return True
    try:
        with open(target + ".srchash") as f:
            return f.read().strip() != _digest(sources, flags)
    except OSError:
        return True


def _stamp(target, sources, flags=""):
    tmp = "%s.srchash.%d.tmp" % (target, os.getpid())
    with open(tmp, "w") as f:
        f.write(_digest(sources, flags) + "\n")
    os.replace(tmp, target + ".srchash")


@contextlib.contextmanager
def _locked(target):
    """One builder at a time per library: the ranks of a torchrun / mp.spawn job all import the package, all see the
    same stale library after a source edit, and must not all run the compiler into the same file while others dlopen it.
    The first rank builds (to a temporary name, moved into place atomically); the rest wait here, re-check and load."""
    with open(target + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _have_compiler(name):
    try:
        return bool(_hipcc() if name == "hipcc" else shutil.which(name))
    except RuntimeError:
        return False


def _keep_prebuilt(target, compiler):
    """A library that is there, on a machine that cannot rebuild it (no compiler): load it as it is, and say so."""
    if os.path.exists(target) and not _have_compiler(compiler):
        sys.stderr.write("[pbr build] %s: no %s here — loading the prebuilt library without checking it against the sources\n" % (os.path.basename(target), compiler))
        return True
    return False


def _run(cmd):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return proc.stdout


HIP_FLAGS = [
    # xnack-: the hand-scheduled node phase lets a record load overwrite its own address register (pt_kernel.hpp,
    # PT_NODE_PHASE_HEAD), which is only legal when a faulted load is never replayed; a plain gfx950 code object is
    # "xnack any" and would also load into an XNACK-enabled process, where a replay would read a clobbered address.
    # With the target feature spelled out the loader enforces the assumption instead.
    "--offload-arch=gfx950:xnack-", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
    # hipcc's SLP vectoriser packs adjacent scalar f32 operations into v_pk_* instructions, which issue at
    # half rate on gfx950 and need v_mov shuffles + hazard s_nops around them: same bits, 3-9 % slower kernels
    "-fno-slp-vectorize",
]


def hip_sources():
    sources = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp")) and os.path.isfile(os.path.join(CSRC, f))]
    return sources + [os.path.join(INCLUDE, f) for f in ("pbr_hip.h", "pbr_hip_diag.h")]


def hip_digest():
    """What the product library is built from: sources + flags.  libpbrhip.so.srchash holds the digest of the build that is
    there; bench.py stamps profiles with it and refuses to price a run with counters of another build."""
    return _digest(hip_sources(), " ".join(HIP_FLAGS + NATIVE_FLAGS))


def build_hip(force=False, guard=False):
    sources = hip_sources()
    flags = " ".join(HIP_FLAGS + NATIVE_FLAGS)
    target = HIP_GUARD_LIB if guard else HIP_LIB
    if not force and (not _stale(target, sources, flags) or _keep_prebuilt(target, "hipcc")):
        return target
    with _locked(target):
        if not force and not _stale(target, sources, flags):     # another process built it while this one waited
            return target
        return _build_hip_locked(target, sources, guard)


# The path-tracing kernels are compiled as one translation unit per (flavour, group) — csrc/pt_instance.hip with
# -DPT_FLAVOUR=f -DPT_GROUP=g, see csrc/pt_instances.hpp / pt_flavour.hpp — next to csrc/pbr_hip.hip (the C ABI, the
# host side of the launches and every other kernel), in parallel, and linked into one shared library.
#   flavour bit 0: ray-ordered walk (pbr_config.traversal)     bit 1: native arithmetic (pbr_config.arith)
#   groups 0-2 pathTracing<.., 4 | 6 | 8>, 3 its Phong-tessellation build (every flavour since round 6), 4-6 pathTracingPhased<.., 4 | 6 | 8>,
#   7 pathTracingDual (not in PBR_GUARD builds: it has no C++ node phase)
#   flavour bit 2 (with bit 0): the compact record of the eight-order walk — flavours 5 and 7 (no two-paths kernel: group 7)
FLAVOURS = (0, 1, 2, 3, 5, 7)
GROUPS = (0, 1, 2, 3, 4, 5, 6, 7)
# what native arithmetic means to the compiler: `/` and sqrtf() become v_rcp_f32 / v_sqrt_f32 sequences without the
# correction steps (the reference asks for native_divide / native_recip / native_sqrt); everything else is in pt_math.hpp
NATIVE_FLAGS = ["-fno-hip-fp32-correctly-rounded-divide-sqrt"]


def instance_units(guard):
    units = []
    for f in FLAVOURS:
        for g in GROUPS:
            if g == 7 and (guard or f & 4):
                continue
            units.append((f, g))
    return units


def hip_commands(target, guard=False, objdir=None):
    """The compile commands (one per translation unit) and the link command of libpbrhip*.so."""
    objdir = objdir or (target + ".obj")
    common = [_hipcc(), *HIP_FLAGS, "-fPIC", "-I", INCLUDE, "-I", CSRC, *(["-DPBR_GUARD=1"] if guard else [])]
    objects = [os.path.join(objdir, "pbr_hip.o")]
    compiles = [common + ["-c", os.path.join(CSRC, "pbr_hip.hip"), "-o", objects[0]]]
    for f, g in instance_units(guard):
        obj = os.path.join(objdir, "inst_f%d_g%d.o" % (f, g))
        objects.append(obj)
        compiles.append(common + (NATIVE_FLAGS if f & 2 else []) + ["-DPT_FLAVOUR=%d" % f, "-DPT_GROUP=%d" % g, "-c", os.path.join(CSRC, "pt_instance.hip"), "-o", obj])
    link = [_hipcc(), "--offload-arch=gfx950:xnack-", "-shared", "-fPIC", *objects, "-o"]
    return compiles, link, objdir


def _build_hip_locked(target, sources, guard):
    from concurrent.futures import ThreadPoolExecutor
    tmp = "%s.%d.tmp" % (target, os.getpid())
    compiles, link, objdir = hip_commands(target, guard, objdir="%s.obj.%d" % (target, os.getpid()))
    os.makedirs(objdir, exist_ok=True)
    try:
        with ThreadPoolExecutor(max_workers=max(1, min(len(compiles), os.cpu_count() or 4))) as pool:
            list(pool.map(_run, compiles))
        _run(link + [tmp])
        os.replace(tmp, target)         # a process that has the old file mapped keeps the old inode
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
        if os.path.exists(tmp):
            os.remove(tmp)
    _stamp(target, sources, " ".join(HIP_FLAGS + NATIVE_FLAGS))
    return target


def build_lab(name, extra_flags=(), flavours=None):
    """A measurement build of the same sources: lab/libpbrhip_<name>.so with -DPBR_LAB_HOOKS -I lab/src (the hooks of
    csrc/pt_kernel.hpp then come from lab/src/pt_lab_hooks.hpp) and the experiment's own -D flags; loaded by the Python
    harness with PBR_LAB_ENV=1 PBR_HIP_LIB=lab/libpbrhip_<name>.so.  Never the product: no digest, not in hip_sources()."""
    from concurrent.futures import ThreadPoolExecutor
    lab = os.path.join(ROOT, "lab")
    target = os.path.join(lab, "libpbrhip_%s.so" % name)
    compiles, link, objdir = hip_commands(target, False, objdir="%s.obj.%d" % (target, os.getpid()))
    extra = ["-DPBR_LAB_HOOKS=1", "-I", os.path.join(lab, "src"), *extra_flags]
    keep = []
    for cmd in compiles:
        flav = [c for c in cmd if c.startswith("-DPT_FLAVOUR=")]
        if flavours is not None and flav and int(flav[0].split("=")[1]) not in flavours:
            continue
        keep.append(cmd[:1] + extra + cmd[1:])
    objects = [cmd[cmd.index("-o") + 1] for cmd in keep]
    os.makedirs(objdir, exist_ok=True)
    try:
        with ThreadPoolExecutor(max_workers=max(1, min(len(keep), os.cpu_count() or 4))) as pool:
            list(pool.map(_run, keep))
        _run([_hipcc(), "--offload-arch=gfx950:xnack-", "-shared", "-fPIC", *objects, "-o", target])
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
    return target


def build_host(force=False):
    sources = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith((".cpp", ".h"))] + [os.path.join(INCLUDE, "pbr_hip.h")]
    build_hip()
    if not force and (not _stale(HOST_LIB, sources) or _keep_prebuilt(HOST_LIB, "g++")):
        return HOST_LIB
    with _locked(HOST_LIB):
        if not force and not _stale(HOST_LIB, sources):
            return HOST_LIB
        return _build_host_locked(sources)


def _build_host_locked(sources):
    tmp = "%s.%d.tmp" % (HOST_LIB, os.getpid())
    cmd = [
        "g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wextra",
        "-I", INCLUDE, "-I", HOST, "-I", "/opt/rocm/include",   # <CL/cl.h> for the cl_* types of host/cl_adaptor.h
        *[os.path.join(HOST, f) for f in HOST_SOURCES],
        "-o", tmp, "-L", CSRC, "-lpbrhip", "-Wl,-rpath,$ORIGIN/../csrc",
    ]
    try:
        _run(cmd)
        os.replace(tmp, HOST_LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    _stamp(HOST_LIB, sources)
    return HOST_LIB


def build_multi(force=False):
    """host/libpbrmulti.so: the in-process multi-GPU driver (host/multi_path_tracer.cpp) — g++ against the HIP runtime API
    and RCCL's headers, linked with libpbrhip, librccl and libamdhip64.  Its own library: only a multi-GPU caller pays for
    loading RCCL."""
    sources = [os.path.join(HOST, f) for f in ("multi_path_tracer.cpp", "multi_path_tracer.h")] + \
        [os.path.join(INCLUDE, f) for f in ("pbr_multi.h", "pbr_hip.h", "pbr_hip_diag.h")]
    build_hip()
    if not force and (not _stale(MULTI_LIB, sources) or _keep_prebuilt(MULTI_LIB, "g++")):
        return MULTI_LIB
    with _locked(MULTI_LIB):
        if not force and not _stale(MULTI_LIB, sources):
            return MULTI_LIB
        tmp = "%s.%d.tmp" % (MULTI_LIB, os.getpid())
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-pthread", "-D__HIP_PLATFORM_AMD__",
               "-I", INCLUDE, "-I", HOST, "-I", "/opt/rocm/include",
               *[os.path.join(HOST, f) for f in MULTI_SOURCES], "-o", tmp,
               "-L", CSRC, "-lpbrhip", "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64",
               "-Wl,-rpath,$ORIGIN/../csrc", "-Wl,-rpath,/opt/rocm/lib"]
        try:
            _run(cmd)
            os.replace(tmp, MULTI_LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        _stamp(MULTI_LIB, sources)
        return MULTI_LIB


def build_all(force=False):
    return build_hip(force), build_host(force), build_multi(force)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--lab":       # python build.py --lab <name> [flags ...]
        print(build_lab(sys.argv[2], [f for arg in sys.argv[3:] for f in arg.split()]))
    else:
        for lib in build_all(force="--force" in sys.argv):
            print(lib)
