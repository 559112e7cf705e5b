// The lane state machine with TWO paths per lane ("phased-dual", plan 6) — included by pt_kernel.hpp inside namespace ptk.
//
// pathTracingPhased hides the latency of a walk's dependent fetches behind six waves per SIMD; this kernel hides it behind
// the lane's OTHER path: every lane carries two path slots A and B (128 paths per wave at 4 waves per SIMD), and the node
// phase of the two walks is software-pipelined — A's next records are requested, then B's records are waited for and tested
// while A's are in flight, and the other way round.  Built in round 4 and measured against the six
// other plans, all bit-identical: 1.02 - 1.03x of phased-mid on the Sponza-class scene (2099 - 2102 against 2048 - 2053 Msamples/s at
// 1080p in 32-frame launches, same box), 0.97x on the Dragon-class scene, 0.89x on the hairball — the tuner takes it where it
// wins (profiles/r04/experiments/two_paths_per_lane.txt).
//
//   node phase   nodePhaseDualPipe (below)
//   leaf phase   POOLED: a lane tests the leaf of whichever of its slots stands on one (A first; the other in the next
//                round) — one pass of the face tests serves lanes of both slots
//   shade phase  POOLED the same way: a lane shades whichever slot waits for shading; the threshold counts lanes
// In registers per slot: ray, 1 / direction, cursor, closest hit, the parked leaf, mode, the walk's counters (17).  Everything
// else a path carries between bounces (PixelState's cold half: 16 dwords) lives in LDS, in four lane-linear 16-byte planes
// per slot behind the staged tree top (P.slotBase; 2 x 64 B x 1024 lanes = 128 KiB, which leaves 1016 staged records), and is
// in registers only while its slot is shaded.  125 VGPRs, no scratch (BRDF 1 without lights).
// Per path the sequence of visits, face tests and random draws is the reference's (pathtracing.cl:207-334, pt_bvh.cl:75-125):
// same image, same debug image, same counters.
#ifdef PT_NODE_PHASE_ASM

// ---- the node phase of two walks, software-pipelined -------------------------------------------------------------------
// Walk A's records live in v[46:53], walk B's in v[64:71], the slab test's temporaries in v54 - v63 (as nodePhaseAsm), the
// cursors in v72 / v73: a prefetched record overwrites v53 / v71.  s[86:87] / s[88:89] = the lanes whose walk A / B goes on,
// s[90:91] / s[92:93] = the lanes that parked on a hit leaf, s80 / s81 = "A's / B's last fetch issued loads of both kinds".
// A wait in front of a slab test is COUNTED — s_waitcnt vmcnt(2) lgkmcnt(2): everything but the other walk's two loads of
// each kind, which were issued later — only when that other fetch did issue both kinds (an instruction with an empty EXEC
// may or may not count); otherwise everything is waited for.  A walk's next records are requested as soon as its visit has
// decided which lanes go on — the exit test runs under the requests; when `keep` or fewer walks go on, each walk takes the visit
// its request is for (a node phase may always run one visit longer) and the phase ends with nothing in flight.
// Per walk 24 vector instructions per visit (nodePhaseAsm: 22) and ~13 scalar ones.  What the loop's shape is worth, measured
// step by step on the Sponza-class scene (profiles/r04/experiments/two_paths_per_lane.txt, 7.): the uncommon cases (a full
// wait, a walk without lanes) out of line, ONE taken branch per iteration instead of three: +1.1 %; everything that does not
// need the records (EXEC, the visit counter) in front of the wait instead of behind it: +0.9 %; the next request issued
// straight from the two masks, its bookkeeping behind the loads, and the parked lanes' leaf words collected once, when the
// phase ends: +0.3 %.  The loop is bound by the length of its dependent chain, not by instruction issue.
#define PT_DUAL_SLAB( n0a, n0b, n0c, n0d, n1a, n1b, oxy, ozz, ixy, izz ) \
		"v_pk_add_f32 v[54:55], v[" n0a ":" n0b "], " oxy " neg_lo:[0,1] neg_hi:[0,1]\n" \
		"v_pk_add_f32 v[56:57], v[" n0c ":" n0d "], " oxy " neg_lo:[0,1] neg_hi:[0,1]\n" \
		"v_pk_add_f32 v[58:59], v[" n1a ":" n1b "], " ozz " neg_lo:[0,1] neg_hi:[0,1]\n" \
		"v_pk_mul_f32 v[54:55], " ixy ", v[54:55]\n" \
		"v_pk_mul_f32 v[56:57], " ixy ", v[56:57]\n" \
		"v_pk_mul_f32 v[58:59], " izz ", v[58:59]\n" \
		"v_min_f32 v60, v54, v56\n" \
		"v_min_f32 v61, v55, v57\n" \
		"v_min_f32 v62, v58, v59\n" \
		"v_max3_f32 v60, v60, v61, v62\n" \
		"v_max_f32 v61, v54, v56\n" \
		"v_max_f32 v63, v58, v59\n" \
		"v_max_f32 v62, v55, v57\n" \
		"v_min3_f32 v61, v61, v62, v63\n"
#define PT_DUAL_FETCH( walk, cur, n0, n1lo, n1hi, flag, skip ) \
		"s_mov_b64 exec, " walk "\n" \
		"s_cbranch_execz " skip "9f\n" \
		"v_cmp_gt_i32 vcc, %[numHotBytes], " cur "\n" \
		"s_and_saveexec_b64 s[94:95], vcc\n" \
		"s_cselect_b32 s82, 1, 0\n" \
		"ds_read_b128 v[" n0 "], " cur "\n" \
		"ds_read_b128 v[" n1lo ":" n1hi "], " cur " offset:16\n" \
		"s_xor_b64 exec, exec, s[94:95]\n" \
		"s_cselect_b32 " flag ", s82, 0\n" \
		"global_load_dwordx4 v[" n0 "], " cur ", %[nodes]\n" \
		"global_load_dwordx4 v[" n1lo ":" n1hi "], " cur ", %[nodes] offset:16\n" \
	skip ":\n"
// the counted wait; the uncommon case (the other fetch did not issue both kinds) waits for everything, out of line
#define PT_DUAL_WAIT( otherFlag, full, go ) \
		"s_cmp_eq_u32 " otherFlag ", 1\n" \
		"s_cbranch_scc0 " full "f\n" \
		"s_waitcnt vmcnt(2) lgkmcnt(2)\n" \
	go ":\n"
// a walk without lanes requests nothing: its flag says so (out of line)
#define PT_DUAL_FETCH_NONE( flag, skip ) \
	skip "9:\n" \
		"s_mov_b32 " flag ", 0\n" \
		"s_branch " skip "b\n"
#define PT_DUAL_WAIT_FULL( full, go ) \
	full ":\n" \
		"s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
		"s_branch " go "b\n"
// One walk's visit: what does not need the records stands in front of the wait; behind it the slab test, the hit chain, and
// s[98:99] = the lanes that stand on a hit leaf.  A parked lane's leaf word stays in the record's register (v52 / v70: no
// later fetch of that walk includes the lane) and is collected when the phase ends; its tNear is a slab temporary, kept at once.
// W = "A" / "B"; walk, park = the walk's mask pairs; r0..r7 = the eight record registers; cur = its cursor register.
#define PT_DUAL_VISIT_CORE( ... ) PT_DUAL_VISIT_CORE_( __VA_ARGS__ )      /* (the register lists are macros: expanded before the call) */
#define PT_DUAL_VISIT_CORE_( skip, wait, W, walk, r0, r1, r2, r3, r4, r5, r6, r7, cur ) \
		"s_mov_b64 exec, " walk "\n" \
		"s_cbranch_execz " skip "f\n" \
		"v_add_u32 %[visits" W "], 1, %[visits" W "]\n" \
		wait \
		PT_DUAL_SLAB( r0, r1, r2, r3, r4, r5, "%[oxy" W "]", "%[ozz" W "]", "%[ixy" W "]", "%[izz" W "]" ) \
		"v_cmpx_lt_f32 %[eps], v61\n" \
		"v_cmpx_gt_f32 %[rayT" W "], v60\n" \
		"v_cmpx_le_f32 v60, v61\n" \
		"v_cmp_gt_i32 s[98:99], 0, v" r6 "\n" \
		"v_cndmask_b32 v" r7 ", v" r6 ", v" r7 ", s[98:99]\n" \
		"v_cndmask_b32 %[tNear" W "], %[tNear" W "], v60, s[98:99]\n" \
		"s_mov_b64 exec, " walk "\n" \
		"v_mov_b32 " cur ", v" r7 "\n"                        /* the cursor, where the next prefetch cannot reach it */ \
		"v_cmp_le_i32 s[94:95], 0, " cur "\n"
// ... and, in the loop, the request for its next records at once: the lanes that go on are EXEC straight from the two masks,
// the bookkeeping (walk mask, parked mask) follows the loads
#define PT_DUAL_VISIT_AND_FETCH( ... ) PT_DUAL_VISIT_AND_FETCH_( __VA_ARGS__ )
#define PT_DUAL_VISIT_AND_FETCH_( skip, wait, none, resume, W, walk, park, flag, r0, r1, r2, r3, r4, r5, r6, r7, cur ) \
		PT_DUAL_VISIT_CORE( skip, wait, W, walk, r0, r1, r2, r3, r4, r5, r6, r7, cur ) \
		"s_andn2_b64 exec, s[94:95], s[98:99]\n" \
		"s_cbranch_scc0 " none "f\n" \
		"v_cmp_gt_i32 vcc, %[numHotBytes], " cur "\n" \
		"s_and_saveexec_b64 s[94:95], vcc\n" \
		"s_cselect_b32 s82, 1, 0\n" \
		"ds_read_b128 v[" r0 ":" r3 "], " cur "\n" \
		"ds_read_b128 v[" r4 ":" r7 "], " cur " offset:16\n" \
		"s_xor_b64 exec, exec, s[94:95]\n" \
		"s_cselect_b32 " flag ", s82, 0\n" \
		"global_load_dwordx4 v[" r0 ":" r3 "], " cur ", %[nodes]\n" \
		"global_load_dwordx4 v[" r4 ":" r7 "], " cur ", %[nodes] offset:16\n" \
		"s_mov_b64 " walk ", s[94:95]\n" \
	resume ":\n" \
		"s_or_b64 " park ", " park ", s[98:99]\n" \
	skip ":\n"
#define PT_DUAL_VISIT_NONE( none, resume, walk, flag ) \
	none ":\n" \
		"s_mov_b32 " flag ", 0\n" \
		"s_mov_b64 " walk ", 0\n" \
		"s_branch " resume "b\n"
#define PT_DUAL_VISIT_LAST( ... ) PT_DUAL_VISIT_LAST_( __VA_ARGS__ )
#define PT_DUAL_VISIT_LAST_( skip, W, walk, park, r0, r1, r2, r3, r4, r5, r6, r7, cur ) \
		PT_DUAL_VISIT_CORE( skip, "s_waitcnt vmcnt(0) lgkmcnt(0)\n", W, walk, r0, r1, r2, r3, r4, r5, r6, r7, cur ) \
		"s_andn2_b64 " walk ", s[94:95], s[98:99]\n" \
		"s_or_b64 " park ", " park ", s[98:99]\n" \
	skip ":\n"
#define PT_DUAL_REGS_A "46", "47", "48", "49", "50", "51", "52", "53", "v72"
#define PT_DUAL_REGS_B "64", "65", "66", "67", "68", "69", "70", "71", "v73"

// refA / refB: the walks' cursors (< 0: this slot sits the phase out).  A lane whose walk parks on a hit leaf gets the
// leaf's word and tNear in leafWordA / tNearA (leafWordB / tNearB); the caller passes 0 in and reads != 0 as "parked".
PT_DEV void nodePhaseDualPipe(
	const DevParams& P,
	const f2v oxyA, const f2v ozzA, const f2v ixyA, const f2v izzA, float rayTA,
	const f2v oxyB, const f2v ozzB, const f2v ixyB, const f2v izzB, float rayTB,
	int keep, int& refA, int& refB, unsigned& visitsA, unsigned& visitsB,
	int& leafWordA, float& tNearA, int& leafWordB, float& tNearB
) {
	const float eps = EPSILON5;
	keep = __builtin_amdgcn_readfirstlane( keep );

	asm volatile(
		"s_waitcnt lgkmcnt(0)\n"
		"s_mov_b64 s[84:85], exec\n"
		"v_cmp_le_i32 s[86:87], 0, %[refA]\n"
		"v_cmp_le_i32 s[88:89], 0, %[refB]\n"
		"s_mov_b64 s[90:91], 0\n"
		"s_mov_b64 s[92:93], 0\n"
		"v_mov_b32 v72, %[refA]\n"
		"v_mov_b32 v73, %[refB]\n"
		PT_DUAL_FETCH( "s[86:87]", "v72", "46:49", "50", "53", "s80", "10" )
		PT_DUAL_FETCH( "s[88:89]", "v73", "64:67", "68", "71", "s81", "11" )
	"1:\n"
		// walk A on its records (B's request, issued after them, may stay in flight), and A's next request; then B the same way
		PT_DUAL_VISIT_AND_FETCH( "4", PT_DUAL_WAIT( "s81", "12", "13" ), "14", "19", "A", "s[86:87]", "s[90:91]", "s80", PT_DUAL_REGS_A )
		PT_DUAL_VISIT_AND_FETCH( "5", PT_DUAL_WAIT( "s80", "15", "16" ), "17", "20", "B", "s[88:89]", "s[92:93]", "s81", PT_DUAL_REGS_B )
		"s_bcnt1_i32_b64 s96, s[86:87]\n"
		"s_bcnt1_i32_b64 s97, s[88:89]\n"
		"s_add_i32 s96, s96, s97\n"
		"s_cmp_gt_i32 s96, %[keep]\n"
		"s_cbranch_scc1 1b\n"
		// enough walks have left.  The records that are on their way are not dropped: each walk takes that visit (a node phase
		// may always run one visit longer), and the phase ends with nothing in flight
		PT_DUAL_VISIT_LAST( "6", "A", "s[86:87]", "s[90:91]", PT_DUAL_REGS_A )
		PT_DUAL_VISIT_LAST( "7", "B", "s[88:89]", "s[92:93]", PT_DUAL_REGS_B )
		"s_waitcnt vmcnt(0) lgkmcnt(0)\n"
		"s_mov_b64 exec, s[90:91]\n"                          // the leaf words of the lanes that parked in this phase
		"v_mov_b32 %[leafWordA], v52\n"
		"s_mov_b64 exec, s[92:93]\n"
		"v_mov_b32 %[leafWordB], v70\n"
		"s_mov_b64 exec, s[84:85]\n"
		"v_mov_b32 %[refA], v72\n"
		"v_mov_b32 %[refB], v73\n"
		"s_branch 21f\n"
		PT_DUAL_WAIT_FULL( "12", "13" )
		PT_DUAL_WAIT_FULL( "15", "16" )
		PT_DUAL_FETCH_NONE( "s80", "10" )
		PT_DUAL_FETCH_NONE( "s81", "11" )
		PT_DUAL_VISIT_NONE( "14", "19", "s[86:87]", "s80" )
		PT_DUAL_VISIT_NONE( "17", "20", "s[88:89]", "s81" )
	"21:\n"
		: [refA] "+v"( refA ), [refB] "+v"( refB ), [visitsA] "+v"( visitsA ), [visitsB] "+v"( visitsB ),
		  [leafWordA] "+v"( leafWordA ), [tNearA] "+v"( tNearA ), [leafWordB] "+v"( leafWordB ), [tNearB] "+v"( tNearB )
		: [oxyA] "v"( oxyA ), [ozzA] "v"( ozzA ), [ixyA] "v"( ixyA ), [izzA] "v"( izzA ), [rayTA] "v"( rayTA ),
		  [oxyB] "v"( oxyB ), [ozzB] "v"( ozzB ), [ixyB] "v"( ixyB ), [izzB] "v"( izzB ), [rayTB] "v"( rayTB ),
		  [keep] "s"( keep ), [numHotBytes] "s"( P.numHotBytes ), [nodes] "s"( P.nodes ), [eps] "s"( eps )
		: "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
		  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73",
		  "s80", "s81", "s82", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "vcc", "scc"
	);
}
#undef PT_DUAL_FETCH
#undef PT_DUAL_WAIT
#undef PT_DUAL_WAIT_FULL
#undef PT_DUAL_FETCH_NONE
#undef PT_DUAL_VISIT_CORE
#undef PT_DUAL_VISIT_CORE_
#undef PT_DUAL_VISIT_AND_FETCH
#undef PT_DUAL_VISIT_AND_FETCH_
#undef PT_DUAL_VISIT_LAST_
#undef PT_DUAL_VISIT_NONE
#undef PT_DUAL_VISIT_LAST
#undef PT_DUAL_REGS_A
#undef PT_DUAL_REGS_B
#undef PT_DUAL_SLAB

// ---- a path slot: what a path keeps in registers while it is not being shaded ----------------------------------------
struct DualSlot {
	Ray ray;
	f3 invDir;
	int cur;            // cursor (byte offset of the next record); < 0: the walk has ended
	float t;            // closest hit so far
	int face;
	int leafWord;       // != 0: parked on this hit leaf (MODE_LEAF)
	float leafTNear;
	int mode;
	unsigned nodes, tris;   // of the current walk(s) since the slot was last shaded: added to the path's counters there
};

// the cold half of a path's state: four lane-linear 16-byte planes per slot in LDS, behind the staged tree top
PT_DEV float4* coldPlane( const DevParams& P, int slot, int plane ) {
	return (float4*) ( (char*) gHotNodes + P.slotBase ) + ( slot * 4 + plane ) * PBR_BLOCK + (int) threadIdx.x;
}

PT_DEV void loadCold( const DevParams& P, int slot, PixelState& st ) {
	const float4 a = *coldPlane( P, slot, 0 ), b = *coldPlane( P, slot, 1 ), c = *coldPlane( P, slot, 2 ), d = *coldPlane( P, slot, 3 );
	st.slot = __float_as_uint( a.x ); st.frame = __float_as_int( a.y ); st.sample = __float_as_int( a.z ); st.finalColor.x = a.w;
	st.finalColor.y = b.x; st.finalColor.z = b.y; st.secondaryPaths = __float_as_uint( b.z ); st.focus = b.w;
	st.seed = c.x; st.dbgNodes = __float_as_uint( c.y ); st.dbgTris = __float_as_uint( c.z ); st.color.x = c.w;
	st.color.y = d.x; st.color.z = d.y; st.depth = __float_as_int( d.z ); st.depthAdded = __float_as_int( d.w );
}

PT_DEV void storeCold( const DevParams& P, int slot, const PixelState& st ) {
	*coldPlane( P, slot, 0 ) = make_float4( __uint_as_float( st.slot ), __int_as_float( st.frame ), __int_as_float( st.sample ), st.finalColor.x );
	*coldPlane( P, slot, 1 ) = make_float4( st.finalColor.y, st.finalColor.z, __uint_as_float( st.secondaryPaths ), st.focus );
	*coldPlane( P, slot, 2 ) = make_float4( st.seed, __uint_as_float( st.dbgNodes ), __uint_as_float( st.dbgTris ), st.color.x );
	*coldPlane( P, slot, 3 ) = make_float4( st.color.y, st.color.z, __int_as_float( st.depth ), __int_as_float( st.depthAdded ) );
}

template<bool LIGHTS>
PT_DEV void startWalkDual( const DevParams& P, DualSlot& s ) {
	s.invDir = mk3( div1( 1.0f, s.ray.dir.x ), div1( 1.0f, s.ray.dir.y ), div1( 1.0f, s.ray.dir.z ) );
	s.cur = firstNode( P, s.ray.dir ).ref;
	Hit h;
	h.t = inff();
	h.face = 0;

	if( LIGHTS ) {
		traverseLights( P, s.ray, h );
	}

	s.t = h.t;
	s.face = h.face;
	s.leafWord = 0;
	s.leafTNear = 0.0f;
	s.mode = MODE_NODE;
}

// P.phPark: a node phase ends once that many of the wave's WALKS (of up to 128) have left it; P.phShade: a shade phase
// runs once that many LANES have a slot waiting (or nothing else is left to do).  Measured best: 28 / 48.
template<int BRDF, bool SHADOW, bool LIGHTS>
__global__ __launch_bounds__( PBR_BLOCK, 4 ) void pathTracingDual( const DevParams P ) {
	const float4* lds = gHotNodes;
	stageHotNodes( P, gHotNodes );
	LaneCounters cnt;
	cnt.nodes = cnt.tris = cnt.hits = cnt.paths = 0;
	DualSlot A, B;
	A.mode = B.mode = MODE_DONE;
	A.cur = B.cur = -1;
	A.leafWord = B.leafWord = 0;
	A.nodes = A.tris = B.nodes = B.tris = 0;
	A.t = B.t = 0.0f;
	A.face = B.face = 0;
	A.leafTNear = B.leafTNear = 0.0f;
	A.ray.origin = A.ray.dir = A.invDir = B.ray.origin = B.ray.dir = B.invDir = mk3( 0.0f, 0.0f, 0.0f );
	WorkCursor work = beginWork();

	// both slots take their first unit
	for( int k = 0; k < 2; k++ ) {
		unsigned frame = 0;
		const unsigned slot = nextSlot( P, work, (unsigned) P.nFrames, frame );

		if( slot != PT_NO_WORK ) {
			PixelState st;
			beginPixel( P, st, slot, cnt, frame );
			DualSlot& S = ( k == 0 ) ? A : B;
			S.ray = st.ray;
			startWalkDual<LIGHTS>( P, S );
			storeCold( P, k, st );
		}
	}

	while( __ballot( A.mode != MODE_DONE || B.mode != MODE_DONE ) != 0ull ) {
		// ---- node phase: both slots of every lane ----------------------------------------------------
		if( A.mode == MODE_NODE || B.mode == MODE_NODE ) {
			const int walking = __popcll( __ballot( A.mode == MODE_NODE ) ) + __popcll( __ballot( B.mode == MODE_NODE ) );
			const int keep = walking - P.phPark;
			const f2v oxyA = { A.ray.origin.x, A.ray.origin.y }, ozzA = { A.ray.origin.z, A.ray.origin.z }, ixyA = { A.invDir.x, A.invDir.y }, izzA = { A.invDir.z, A.invDir.z };
			const f2v oxyB = { B.ray.origin.x, B.ray.origin.y }, ozzB = { B.ray.origin.z, B.ray.origin.z }, ixyB = { B.invDir.x, B.invDir.y }, izzB = { B.invDir.z, B.invDir.z };
			int refA = ( A.mode == MODE_NODE ) ? A.cur : -1;      // a slot that is not walking sits the phase out
			int refB = ( B.mode == MODE_NODE ) ? B.cur : -1;
			int leafWordA = 0, leafWordB = 0;
			float tNearA = 0.0f, tNearB = 0.0f;
			__builtin_amdgcn_s_setprio( PT_WALK_PRIO );
			nodePhaseDualPipe( P, oxyA, ozzA, ixyA, izzA, A.t, oxyB, ozzB, ixyB, izzB, B.t, ( keep < 0 ) ? 0 : keep, refA, refB, A.nodes, B.nodes,
			                   leafWordA, tNearA, leafWordB, tNearB );
			__builtin_amdgcn_s_setprio( 0 );

			if( A.mode == MODE_NODE ) {
				A.cur = refA;
				A.leafWord = leafWordA;
				A.leafTNear = tNearA;
				A.mode = ( leafWordA != 0 ) ? MODE_LEAF : ( ( refA < 0 ) ? MODE_SHADE : MODE_NODE );
			}
			if( B.mode == MODE_NODE ) {
				B.cur = refB;
				B.leafWord = leafWordB;
				B.leafTNear = tNearB;
				B.mode = ( leafWordB != 0 ) ? MODE_LEAF : ( ( refB < 0 ) ? MODE_SHADE : MODE_NODE );
			}
		}

		// ---- leaf phase, pooled: the slot that stands on a leaf (A first; B's in the next round) ------------
		{
			const bool leafA = ( A.mode == MODE_LEAF ), leafB = ( B.mode == MODE_LEAF );

			if( leafA || leafB ) {
				const bool useB = !leafA;
				Ray ray;
				ray.origin = useB ? B.ray.origin : A.ray.origin;
				ray.dir = useB ? B.ray.dir : A.ray.dir;
				Hit hit;
				hit.t = useB ? B.t : A.t;
				hit.face = useB ? B.face : A.face;
				const int leafWord = useB ? B.leafWord : A.leafWord;
				const float tNear = useB ? B.leafTNear : A.leafTNear;
				unsigned tests = 0;
				__builtin_amdgcn_s_setprio( PT_WALK_PRIO );
				testLeaf<false, true>( P, leafFace0( leafWord ), leafFace1( leafWord ), ray, tNear, 0.0f, hit, tests );
				__builtin_amdgcn_s_setprio( 0 );

				if( useB ) {
					B.t = hit.t; B.face = hit.face; B.tris += tests; B.leafWord = 0;
					B.mode = ( B.cur < 0 ) ? MODE_SHADE : MODE_NODE;
				}
				else {
					A.t = hit.t; A.face = hit.face; A.tris += tests; A.leafWord = 0;
					A.mode = ( A.cur < 0 ) ? MODE_SHADE : MODE_NODE;
				}
			}
		}

		// ---- shade phase, pooled: the slot that waits for shading (A first); the threshold counts lanes ------
		{
			const bool shadeA = ( A.mode == MODE_SHADE ), shadeB = ( B.mode == MODE_SHADE );
			const int nShade = __popcll( __ballot( shadeA || shadeB ) );
			const bool busy = ( __ballot( A.mode == MODE_NODE || A.mode == MODE_LEAF || B.mode == MODE_NODE || B.mode == MODE_LEAF ) != 0ull );

			if( ( shadeA || shadeB ) && ( nShade >= P.phShade || !busy ) ) {
				const bool useB = !shadeA;
				const int which = useB ? 1 : 0;
				PixelState st;
				loadCold( P, which, st );
				st.ray.origin = useB ? B.ray.origin : A.ray.origin;
				st.ray.dir = useB ? B.ray.dir : A.ray.dir;
				st.dbgNodes += useB ? B.nodes : A.nodes;
				st.dbgTris += useB ? B.tris : A.tris;
				Hit hit;
				hit.t = useB ? B.t : A.t;
				hit.face = useB ? B.face : A.face;
				hit.normal = mk3( 0.0f, 0.0f, 0.0f );
				bool more = true;

				if( shadeStep<BRDF, SHADOW, LIGHTS, false, true, true>( P, lds, st, cnt, hit ) ) {
					finishPixel( P, st );

					if( cnt.nodes > 0x40000000u || cnt.tris > 0x40000000u ) {
						flushCounters( P, cnt );
					}

					unsigned frame = 0;
					const unsigned slot = nextSlot( P, work, (unsigned) P.nFrames, frame );

					if( slot != PT_NO_WORK ) {
						beginPixel( P, st, slot, cnt, frame );
					}
					else {
						more = false;
					}
				}

				storeCold( P, which, st );

				if( useB ) {
					B.nodes = 0; B.tris = 0; B.ray = st.ray;
					if( more ) { startWalkDual<LIGHTS>( P, B ); } else { B.mode = MODE_DONE; B.cur = -1; }
				}
				else {
					A.nodes = 0; A.tris = 0; A.ray = st.ray;
					if( more ) { startWalkDual<LIGHTS>( P, A ); } else { A.mode = MODE_DONE; A.cur = -1; }
				}
			}
		}
	}

	flushCounters( P, cnt );
}
#endif   // PT_NODE_PHASE_ASM
