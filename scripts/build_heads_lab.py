# lab: the library (default flavour) with N queue heads, N/8 per XCD, from a patched COPY of csrc -> lab/libpbrhip_headsN.so
# (scripts/heads_ab.sh runs it against the product).  usage: python scripts/build_heads_lab.py 16
import importlib.util, os, shutil, sys
spec = importlib.util.spec_from_file_location("b", "/root/repo/physically-based-rendering_amd/build.py")
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
heads = int(sys.argv[1])
src = "/tmp/labsrc%d" % heads
shutil.rmtree(src, ignore_errors=True)
shutil.copytree(b.CSRC, src, ignore=shutil.ignore_patterns("*.so", "*.srchash", "*.lock", "*.obj*"))
p = os.path.join(src, "pt_kernel.hpp")
s = open(p).read()
old = "wc.home = (int) ( xcc & ( PT_BANDS - 1 ) );"
assert old in s
s = s.replace(old, "wc.home = (int) ( ( ( xcc & 7u ) * ( PT_BANDS / 8 ) + ( ( threadIdx.x >> 6 ) % ( PT_BANDS / 8 ) ) ) & ( PT_BANDS - 1 ) );")
open(p, "w").write(s)
b.CSRC = src
print(b.build_lab("heads%d" % heads, ["-DPT_BANDS=%d" % heads], flavours=[0]))
