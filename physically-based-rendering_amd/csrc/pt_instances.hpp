// Where the path-tracing kernels are instantiated: one translation unit per (flavour, group), all from pt_instance.hip
// (build.py compiles it once per pair with -DPT_FLAVOUR=f -DPT_GROUP=g, in parallel, and links the objects into
// libpbrhip.so).  A group is one plan's kernel in its six variants (BRDF 0 / 1 x no lights / lights / lights + shadow rays);
// a flavour is a build mode of the same sources (pt_flavour.hpp).  Each unit exports ONE function that picks the variant.
#pragma once

#include <stdint.h>

#define PTI_FLAVOURS 8   // bit 0: ray-ordered walk, bit 1: native arithmetic, bit 2: compact record of the eight-order walk (only with bit 0: flavours 5 and 7)
#define PTI_REFILL_LEAN 0    // pathTracing<.., 4>            lock step per bounce, <= 128 VGPRs
#define PTI_REFILL_MID 1     // pathTracing<.., 6>            <= 80
#define PTI_REFILL_WIDE 2    // pathTracing<.., 8>            <= 64
#define PTI_REFILL_PHONG 3   // pathTracing<.., 4, PHONG>     Phong tessellation
#define PTI_PHASED_LEAN 4    // pathTracingPhased<.., 4>      lane state machine
#define PTI_PHASED_MID 5     // pathTracingPhased<.., 6>
#define PTI_PHASED_WIDE 6    // pathTracingPhased<.., 8>
#define PTI_DUAL 7           // pathTracingDual               two paths per lane (not in PBR_GUARD builds)
#define PTI_GROUPS 8

// the address of the kernel's host stub (a void (*)( const DevParams ) of that flavour's namespace; all flavours share
// DevParams' layout), or NULL
typedef const void* ( *pti_picker )( uint32_t brdf, int shadow, int lights );

#define PTI_NAME2( f, g ) pti_pick_f##f##_g##g
#define PTI_NAME( f, g ) PTI_NAME2( f, g )
