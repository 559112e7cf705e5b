# (patches the csrc of commit d8000ff — one queue head per band; later trees have the sub-heads built in and the patch strings no longer match)
# lab: the library (default flavour) with N queue heads, N/8 per XCD, from a patched COPY of csrc -> lab/libpbrhip_headsN[g].so
# (scripts/heads_ab.sh runs it against the product).
#   python scripts/build_heads_lab.py 16            two heads per XCD, a block's waves alternate; thieves walk the heads in cyclic order
#   python scripts/build_heads_lab.py 32 grouped    four per XCD; a wave that steals goes XCD group by XCD group and, inside a group,
#                                                   starts at ITS OWN sub-head: the thieves of one band spread over its four heads
import importlib.util, os, shutil, sys
spec = importlib.util.spec_from_file_location("b", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "physically-based-rendering_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
heads = int(sys.argv[1])
grouped = len(sys.argv) > 2 and sys.argv[2] == "grouped"
src = "/tmp/labsrc%d%s" % (heads, "g" if grouped else "")
shutil.rmtree(src, ignore_errors=True)
shutil.copytree(b.CSRC, src, ignore=shutil.ignore_patterns("*.so", "*.srchash", "*.lock", "*.obj*"))
p = os.path.join(src, "pt_kernel.hpp")
s = open(p).read()


def swap(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new)


swap("wc.home = (int) ( xcc & ( PT_BANDS - 1 ) );",
     "wc.home = (int) ( ( ( xcc & 7u ) * ( PT_BANDS / 8 ) + ( ( threadIdx.x >> 6 ) % ( PT_BANDS / 8 ) ) ) & ( PT_BANDS - 1 ) );")
if grouped:
    # `exhausted` is kept in the wave's own visiting order: position p = group step * SUB + sub step
    swap("while( wc.exhausted != ( 1u << PT_BANDS ) - 1u ) {", "while( wc.exhausted != (unsigned) ( ( 1ull << PT_BANDS ) - 1ull ) ) {")
    swap("""		const unsigned rotated = ( ( wc.exhausted >> wc.home ) | ( wc.exhausted << ( PT_BANDS - wc.home ) ) ) & ( ( 1u << PT_BANDS ) - 1u );
		const int mine = ( wc.home + __builtin_ctz( ~rotated ) ) & ( PT_BANDS - 1 );""",
         """		constexpr unsigned SUB = PT_BANDS / 8;
		const unsigned pos = (unsigned) __builtin_ctz( ~wc.exhausted );
		const unsigned homeGroup = (unsigned) wc.home / SUB, homeSub = (unsigned) wc.home % SUB;
		const int mine = (int) ( ( ( homeGroup + pos / SUB ) & 7u ) * SUB + ( ( homeSub + pos % SUB ) % SUB ) );""")
    swap("			wc.exhausted |= 1u << band;",
         "			wc.exhausted |= 1u << ( ( ( (unsigned) band / SUB - homeGroup ) & 7u ) * SUB + ( ( (unsigned) band % SUB + SUB - homeSub ) % SUB ) );")
open(p, "w").write(s)
b.CSRC = src
print(b.build_lab("heads%d%s" % (heads, "g" if grouped else ""), ["-DPT_BANDS=%d" % heads], flavours=[0]))
