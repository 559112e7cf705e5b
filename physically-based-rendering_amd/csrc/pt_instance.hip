// One group of path-tracing kernels in one build flavour (pt_instances.hpp, pt_flavour.hpp):
//   hipcc -c -DPT_FLAVOUR=<0..3 | 5 | 7> -DPT_GROUP=<0..7> pt_instance.hip
#if !defined( PT_FLAVOUR ) || !defined( PT_GROUP )
#error "pt_instance.hip is compiled once per (PT_FLAVOUR, PT_GROUP) pair: see build.py"
#endif

#include "pt_kernel.hpp"
#include "pt_instances.hpp"

#if PT_GROUP == PTI_DUAL && PT_WALK_COMPACT == 1
#error "pathTracingDual has no node phase for the compact record: the compact flavours render plan 6 with the 6-waves state machine"
#endif
#if PT_GROUP == PTI_DUAL && !defined( PT_NODE_PHASE_ASM )
#error "pathTracingDual has no C++ node phase: PBR_GUARD builds render plan 6 with the 6-waves state machine"
#endif

namespace {

#if PT_GROUP == PTI_REFILL_LEAN
#define PTI_KERNEL( B, S, L ) ptk::pathTracing<B, S, L, 4, false>
#elif PT_GROUP == PTI_REFILL_MID
#define PTI_KERNEL( B, S, L ) ptk::pathTracing<B, S, L, 6, false>
#elif PT_GROUP == PTI_REFILL_WIDE
#define PTI_KERNEL( B, S, L ) ptk::pathTracing<B, S, L, 8, false>
#elif PT_GROUP == PTI_REFILL_PHONG
#define PTI_KERNEL( B, S, L ) ptk::pathTracing<B, S, L, 4, true>
#elif PT_GROUP == PTI_PHASED_LEAN
#define PTI_KERNEL( B, S, L ) ptk::pathTracingPhased<B, S, L, 4>
#elif PT_GROUP == PTI_PHASED_MID
#define PTI_KERNEL( B, S, L ) ptk::pathTracingPhased<B, S, L, 6>
#elif PT_GROUP == PTI_PHASED_WIDE
#define PTI_KERNEL( B, S, L ) ptk::pathTracingPhased<B, S, L, 8>
#elif PT_GROUP == PTI_DUAL
#define PTI_KERNEL( B, S, L ) ptk::pathTracingDual<B, S, L>
#else
#error "unknown PT_GROUP"
#endif

typedef void ( *Kernel )( const ptk::DevParams );

}  // namespace

extern "C" const void* PTI_NAME( PT_FLAVOUR, PT_GROUP )( uint32_t brdf, int shadow, int lights ) {
	Kernel k;

	if( brdf == 0 ) {
		k = lights ? ( shadow ? PTI_KERNEL( 0, true, true ) : PTI_KERNEL( 0, false, true ) ) : PTI_KERNEL( 0, false, false );
	}
	else {
		k = lights ? ( shadow ? PTI_KERNEL( 1, true, true ) : PTI_KERNEL( 1, false, true ) ) : PTI_KERNEL( 1, false, false );
	}

	return (const void*) k;
}
