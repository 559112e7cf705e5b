# (patches the csrc of commit 2e5f632 — four heads per band, nothing published; the product has had this since 2f234ff)
# lab: the product's queue + a PUBLISHED mask of the heads that have been seen empty (one word behind the heads): a wave that finds
# a head empty ORs its bit in and takes what the others have published from the returned value — the end of a launch costs a wave
# two round trips instead of one per head (32).  From a patched COPY of csrc -> lab/libpbrhip_published.so
import importlib.util, os, shutil, sys
spec = importlib.util.spec_from_file_location("b", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "physically-based-rendering_amd", "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
src = "/tmp/labsrc_published"
shutil.rmtree(src, ignore_errors=True)
shutil.copytree(b.CSRC, src, ignore=shutil.ignore_patterns("*.so", "*.srchash", "*.lock", "*.obj*"))


def patch(name, pairs):
    p = os.path.join(src, name)
    s = open(p).read()
    for old, new in pairs:
        assert old in s, old
        s = s.replace(old, new)
    open(p, "w").write(s)


patch("pt_kernel.hpp", [
    ("""			wc.exhausted |= 1u << ( ( ( (unsigned) band - homeBand ) & ( PT_BANDS - 1 ) ) * PT_SUB + ( ( sub - homeSub ) & ( PT_SUB - 1 ) ) );
			continue;""",
     """			wc.exhausted |= 1u << ( ( ( (unsigned) band - homeBand ) & ( PT_BANDS - 1 ) ) * PT_SUB + ( ( sub - homeSub ) & ( PT_SUB - 1 ) ) );
			// publish it, and take what the other waves have published (bit h of the word behind the heads: head h is empty)
			const unsigned seen = (unsigned) __builtin_amdgcn_readfirstlane( (int) ( atomicOr( P.workCounter + PT_HEADS * PT_BAND_STRIDE, 1u << head ) ) );
			const unsigned byBand = __funnelshift_r( seen, seen, homeBand * PT_SUB );      // nibble g = band homeBand + g
			const unsigned low = 0x11111111u * ( 0xFu >> homeSub ), high = 0x11111111u * ( ( 0xFu << ( PT_SUB - homeSub ) ) & 0xFu );
			wc.exhausted |= ( ( byBand >> homeSub ) & low ) | ( ( byBand << ( PT_SUB - homeSub ) ) & high );
			continue;"""),
])
patch("pt_aux.hpp", [("if( slot < (unsigned) PT_HEADS ) {", "if( slot <= (unsigned) PT_HEADS ) {")])
patch("pbr_hip.hip", [("sizeof( unsigned int ) * PT_HEADS * PT_BAND_STRIDE;", "sizeof( unsigned int ) * ( PT_HEADS + 1 ) * PT_BAND_STRIDE;")])
b.CSRC = src
print(b.build_lab("published", [], flavours=[0]))
