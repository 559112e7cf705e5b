// Build flavours of the path-tracing kernels.
//
// The kernels of pt_kernel.hpp / pt_dual.hpp are sized to the register: the 6-waves state machine has 80 registers and no
// scratch, and "anything next to its loop is paid for inside it" (DESIGN.md section 9).  Two opt-in modes of round 5 —
// the ray-ordered walk (pbr_config.traversal) and the native arithmetic (pbr_config.arith) — would each put code next to
// that loop: as run-time branches they cost the reference mode 2 - 8 registers per kernel and brought spills back into the
// 80-register builds (measured: 16 B of scratch in pathTracingPhased<1, false, false, 6>).  So they are BUILD flavours: the
// same sources compiled again with PT_FLAVOUR set, each flavour in its own namespace (the kernels of two flavours have the
// same template arguments and must not share a symbol), each plan of each flavour in its own translation unit
// (pt_instance.hip, build.py) — which also lets the build run on all cores.
//
//   PT_FLAVOUR  bit 0  ray-ordered walk: a walk starts at the first record of the ray's order (firstNode)
//               bit 1  native arithmetic: v_rcp / v_sqrt / v_sin / v_cos / v_log / v_exp instead of the exact definitions
//               bit 2  (with bit 0, round 6) the COMPACT record of the eight-order walk: one 64-byte record per node shared by
//                      the eight orders instead of eight streams of 32-byte records (pt_kernel.hpp, "the compact record")
//   undefined          the translation unit of pbr_hip.hip: no path-tracing kernel is instantiated there; its diagnostic and
//                      denoise kernels take the walk from DevParams.walkScheme at run time and compute exactly
#pragma once

#define PT_CAT2( a, b ) a##b
#define PT_CAT( a, b ) PT_CAT2( a, b )

#ifdef PT_FLAVOUR
#define ptk PT_CAT( ptk_f, PT_FLAVOUR )
#define ptm PT_CAT( ptm_f, PT_FLAVOUR )
#define PT_WALK_MODE ( PT_FLAVOUR & 1 )           // 0: the reference's order only; 1: a ray-ordered walk only
#define PT_ARITH_NATIVE ( ( PT_FLAVOUR >> 1 ) & 1 )
#define PT_WALK_COMPACT ( ( PT_FLAVOUR >> 2 ) & 1 )   // 1: the node stream holds compact 64-byte records (walkScheme 3)
#else
#define PT_WALK_MODE 2                             // decided per launch (DevParams.walkScheme)
#define PT_ARITH_NATIVE 0
#define PT_WALK_COMPACT 2                          // decided per launch (DevParams.walkScheme == 3)
#endif
