# (patches the csrc of commit d8000ff — one queue head per band; later trees have the sub-heads built in and the patch strings no longer match)
# lab: the product candidate of the sub-head queue, from a patched COPY of csrc -> lab/libpbrhip_sub<S>.so.  The eight bands and
# their tables stay as they are; every band gets S heads, head s deals the band's tiles s, s + S, s + 2 S ... (of its order, whatever it
# is); a wave's home is (its XCD's band, its wave index mod S); a wave that steals goes band by band and starts, inside a band, at its
# own sub-head.   usage: python scripts/build_subheads_lab.py 4 | 8
import importlib.util, os, shutil, sys
spec = importlib.util.spec_from_file_location("b", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "physically-based-rendering_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
sub = int(sys.argv[1])
src = "/tmp/labsrc_sub%d" % sub
shutil.rmtree(src, ignore_errors=True)
shutil.copytree(b.CSRC, src, ignore=shutil.ignore_patterns("*.so", "*.srchash", "*.lock", "*.obj*"))


def patch(name, pairs):
    p = os.path.join(src, name)
    s = open(p).read()
    for old, new in pairs:
        assert old in s, old
        s = s.replace(old, new)
    open(p, "w").write(s)


mask_t = "unsigned long long" if sub * 8 > 32 else "unsigned"
one = "1ull" if sub * 8 > 32 else "1u"
ctz = "__builtin_ctzll" if sub * 8 > 32 else "__builtin_ctz"
full = "~0ull" if sub * 8 == 64 else ("~0u" if sub * 8 == 32 else "( ( 1u << PT_HEADS ) - 1u )")
patch("pt_kernel.hpp", [
    ("#define PT_BAND_STRIDE 32", "#define PT_SUB %d\n#define PT_HEADS ( PT_BANDS * PT_SUB )\n#define PT_BAND_STRIDE 32" % sub),
    ("	unsigned exhausted;   // bit b: this lane has seen band b empty", "	%s exhausted;   // bit p: this lane has seen the p-th head of its visiting order empty" % mask_t),
    ("wc.home = (int) ( xcc & ( PT_BANDS - 1 ) );", "wc.home = (int) ( ( xcc & ( PT_BANDS - 1 ) ) * PT_SUB + ( (unsigned) __builtin_amdgcn_readfirstlane( (int) ( threadIdx.x >> 6 ) ) & ( PT_SUB - 1 ) ) );"),
    ("while( wc.exhausted != ( 1u << PT_BANDS ) - 1u ) {", "while( wc.exhausted != %s ) {" % full),
    ("""		const unsigned rotated = ( ( wc.exhausted >> wc.home ) | ( wc.exhausted << ( PT_BANDS - wc.home ) ) ) & ( ( 1u << PT_BANDS ) - 1u );
		const int mine = ( wc.home + __builtin_ctz( ~rotated ) ) & ( PT_BANDS - 1 );
		const int band = __builtin_amdgcn_readfirstlane( mine );
		const unsigned q = atomicAdd( P.workCounter + band * PT_BAND_STRIDE, 1u );""",
     """		const unsigned pos = (unsigned) %s( ~wc.exhausted );
		const unsigned homeBand = (unsigned) wc.home / PT_SUB, homeSub = (unsigned) wc.home & ( PT_SUB - 1 );
		const int mine = (int) ( ( ( homeBand + pos / PT_SUB ) & ( PT_BANDS - 1 ) ) * PT_SUB + ( ( homeSub + pos ) & ( PT_SUB - 1 ) ) );
		const int head = __builtin_amdgcn_readfirstlane( mine );
		const int band = head / PT_SUB;
		const unsigned sub = (unsigned) head & ( PT_SUB - 1 );
		const unsigned q = atomicAdd( P.workCounter + head * PT_BAND_STRIDE, 1u );""" % ctz),
    ("		const unsigned bandSlots = P.bandTiles[band] * 64u;", "		const unsigned bandSlots = ( ( P.bandTiles[band] + ( PT_SUB - 1 ) - sub ) / PT_SUB ) * 64u;"),
    ("			wc.exhausted |= 1u << band;", "			wc.exhausted |= %s << ( ( ( (unsigned) band - homeBand ) & ( PT_BANDS - 1 ) ) * PT_SUB + ( ( sub - homeSub ) & ( PT_SUB - 1 ) ) );" % one),
    ("( ( P.bandFirst[band] + ( qf >> 6 ) ) << 2 ) );", "( ( P.bandFirst[band] + ( qf >> 6 ) * PT_SUB + sub ) << 2 ) );"),
])
patch("pt_aux.hpp", [("if( slot < (unsigned) PT_BANDS ) {", "if( slot < (unsigned) PT_HEADS ) {")])
patch("pbr_hip.hip", [("sizeof( unsigned int ) * PT_BANDS * PT_BAND_STRIDE;", "sizeof( unsigned int ) * PT_HEADS * PT_BAND_STRIDE;")])
b.CSRC = src
print(b.build_lab("sub%d" % sub, [], flavours=[0]))
