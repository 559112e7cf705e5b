// Round-4 lab variants (NOT part of the product build: included by csrc/pt_kernel.hpp only under -DPBR_LAB, scripts/lab.sh).
// The node-phase variants: nodePhasePair (adjacent record fetched along), nodePhaseDual (two walks per lane), nodePhaseAsync (polled LDS-DMA slots).
// Measured in profiles/r04/experiments/; DESIGN.md section 5.1e says what each was for and why the product does not use it.

// ---- the node phase with the adjacent record fetched along (round 4, lab) ----------------------------------
// VERDICT r03 item 1.  A cold fetch brings 64 B: the record and the next one of the stream (in DFS order a container's hit
// successor; the same 128-byte line three times out of four).  pbr_upload_scene keeps a second copy of the stream for
// this plan in which a container's w0 carries bit 0 when (a) the record lies behind the ranked prefix — so every lane that
// sees it has fetched from memory, whatever the plan's LDS share — and (b) its hit successor is the adjacent record.  A
// lane whose box is hit and whose new cursor carries the flag takes its NEXT visit at once, from registers: same
// visit, same counters, one round trip to memory less.  Per lane the sequence of visits is the reference's
// (pt_bvh.cl:88-122).  Registers: as nodePhaseAsm + v[64:71] for the adjacent record.
template<int DUMMY = 0>
PT_DEV void nodePhasePair(
	const DevParams& P, const f2v oxy, const f2v ozz, const f2v ixy, const f2v izz, float rayT, int keep,
	int& ref, unsigned& visits, int& leafWord, float& leafTNear, int& parked
) {
	const float eps = EPSILON5;
	keep = __builtin_amdgcn_readfirstlane( keep );
	unsigned long long saved, active, parkMask, mA;
	int count;

	asm volatile(
		"s_mov_b64 %[saved], exec\n"
		"s_mov_b64 %[parkMask], 0\n"
		"v_mov_b32 v53, %[ref]\n"
	"1:\n"
		"v_cmp_gt_i32 vcc, %[numHotBytes], v53\n"
		"s_and_saveexec_b64 %[active], vcc\n"
		"ds_read_b128 v[46:49], v53\n"
		"ds_read_b128 v[50:53], v53 offset:16\n"
		"s_xor_b64 exec, exec, %[active]\n"
		"global_load_dwordx4 v[46:49], v53, %[nodes]\n"
		"global_load_dwordx4 v[64:67], v53, %[nodes] offset:32\n"
		"global_load_dwordx4 v[68:71], v53, %[nodes] offset:48\n"
		"global_load_dwordx4 v[50:53], v53, %[nodes] offset:16\n"
		"s_mov_b64 exec, %[active]\n"
		"v_add_u32 %[visits], 1, %[visits]\n"
		"s_waitcnt vmcnt(0) lgkmcnt(0)\n"
		"v_pk_add_f32 v[54:55], v[46:47], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_add_f32 v[56:57], v[48:49], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_add_f32 v[58:59], v[50:51], %[ozz] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_mul_f32 v[54:55], %[ixy], v[54:55]\n"
		"v_pk_mul_f32 v[56:57], %[ixy], v[56:57]\n"
		"v_pk_mul_f32 v[58:59], %[izz], v[58:59]\n"
		"v_min_f32 v60, v54, v56\n"
		"v_min_f32 v61, v55, v57\n"
		"v_min_f32 v62, v58, v59\n"
		"v_max3_f32 v60, v60, v61, v62\n"
		"v_max_f32 v61, v54, v56\n"
		"v_max_f32 v63, v58, v59\n"
		"v_max_f32 v62, v55, v57\n"
		"v_min3_f32 v61, v61, v62, v63\n"
		"v_cmpx_lt_f32 %[eps], v61\n"
		"v_cmpx_gt_f32 %[rayT], v60\n"
		"v_cmpx_le_f32 v60, v61\n"
		"v_cmp_gt_i32 vcc, 0, v52\n"
		"v_cndmask_b32 v53, v52, v53, vcc\n"
		"s_or_b64 %[parkMask], %[parkMask], vcc\n"
		// the lanes whose next record is the adjacent one they have fetched: their next visit, from registers
		"v_and_b32 v62, 0x80000001, v53\n"                   // (a walk that has ended is all ones: not a flag)
		"v_cmp_eq_u32 vcc, 1, v62\n"
		"s_and_b64 exec, exec, vcc\n"
		"s_cbranch_scc0 2f\n"
		"s_mov_b64 %[mA], exec\n"
		"v_add_u32 %[visits], 1, %[visits]\n"
		"v_pk_add_f32 v[54:55], v[64:65], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_add_f32 v[56:57], v[66:67], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_add_f32 v[58:59], v[68:69], %[ozz] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_mul_f32 v[54:55], %[ixy], v[54:55]\n"
		"v_pk_mul_f32 v[56:57], %[ixy], v[56:57]\n"
		"v_pk_mul_f32 v[58:59], %[izz], v[58:59]\n"
		"v_min_f32 v60, v54, v56\n"
		"v_min_f32 v61, v55, v57\n"
		"v_min_f32 v62, v58, v59\n"
		"v_max3_f32 v60, v60, v61, v62\n"
		"v_max_f32 v61, v54, v56\n"
		"v_max_f32 v63, v58, v59\n"
		"v_max_f32 v62, v55, v57\n"
		"v_min3_f32 v61, v61, v62, v63\n"
		"v_cmpx_lt_f32 %[eps], v61\n"
		"v_cmpx_gt_f32 %[rayT], v60\n"
		"v_cmpx_le_f32 v60, v61\n"
		"v_cmp_gt_i32 vcc, 0, v70\n"
		"v_cndmask_b32 v71, v70, v71, vcc\n"
		"s_or_b64 %[parkMask], %[parkMask], vcc\n"
		"v_mov_b32 v52, v70\n"                               // a lane that parks here: its leaf word where the epilogue reads it
		"s_mov_b64 exec, %[mA]\n"
		"v_and_b32 v53, -2, v71\n"                           // the cursor never carries the flag into an address
	"2:\n"
		"s_mov_b64 exec, %[active]\n"
		"v_cmp_le_i32 %[mA], 0, v53\n"
		"s_andn2_b64 exec, %[mA], %[parkMask]\n"
		"s_bcnt1_i32_b64 %[count], exec\n"
		"s_cmp_gt_i32 %[count], %[keep]\n"
		"s_cbranch_scc1 1b\n"
		"s_mov_b64 exec, %[saved]\n"
		"v_mov_b32 %[ref], v53\n"
		"v_cndmask_b32 %[parked], 0, 1, %[parkMask]\n"
		"v_mov_b32 %[leafWord], v52\n"
		"v_mov_b32 %[leafTNear], v60\n"
		: [ref] "+v"( ref ), [visits] "+v"( visits ), [leafWord] "=v"( leafWord ), [leafTNear] "=v"( leafTNear ), [parked] "=v"( parked ),
		  [saved] "=&s"( saved ), [active] "=&s"( active ), [parkMask] "=&s"( parkMask ), [mA] "=&s"( mA ), [count] "=&s"( count )
		: [oxy] "v"( oxy ), [ozz] "v"( ozz ), [ixy] "v"( ixy ), [izz] "v"( izz ), [rayT] "v"( rayT ), [keep] "s"( keep ),
		  [numHotBytes] "s"( P.numHotBytes ), [nodes] "s"( P.nodes ), [eps] "s"( eps )
		: "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
		  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "vcc", "scc"
	);
}

// ---- the node phase with PAIRED cold fetches (round 4, lab; -DPBR_HALVES with -DPBR_PAIR_LEAN) ----------------------
// scripts/micro/pair_coalesce.hip: the L1 / TA charges a divergent global_load_dwordx4 partly per distinct 64-byte segment.
// Here the two lanes of a pair (2k, 2k + 1) fetch the two 16-byte halves of ONE record with one instruction — first the
// even lane's record (if it is cold), then the odd lane's: every load instruction touches one segment per pair instead of
// one per lane.  Afterwards a lane takes the half its partner fetched for it (DPP quad_perm [1,0,3,2]) and its own from
// the pair registers; lanes on staged records and lanes that do not walk only lend their load slots (their own registers
// v46 - v63 are not written: the selects keep them).  Per lane the sequence of visits is the reference's; +20 vector and
// +13 scalar instructions per iteration.  Registers: as nodePhaseAsm + v[64:71] (the two paired loads), v72 / v73.
template<int DUMMY = 0>
PT_DEV void nodePhaseHalves(
	const DevParams& P, const f2v oxy, const f2v ozz, const f2v ixy, const f2v izz, float rayT, int keep,
	int& ref, unsigned& visits, int& leafWord, float& leafTNear, int& parked
) {
	const float eps = EPSILON5;
	keep = __builtin_amdgcn_readfirstlane( keep );
	unsigned long long saved, active, parkMask, mA;
	int count;
	const int laneHalf = ( (int) threadIdx.x & 1 ) * 16;     // which half of a record this lane fetches

	asm volatile(
		"s_mov_b64 %[saved], exec\n"
		"s_mov_b64 %[parkMask], 0\n"
		"v_mov_b32 v53, %[ref]\n"
		"v_mov_b32 v72, %[laneHalf]\n"
		"s_mov_b32 s80, 0x55555555\n"
		"s_mov_b32 s81, 0x55555555\n"
	"1:\n"
		"v_cmp_gt_i32 vcc, %[numHotBytes], v53\n"
		"s_and_saveexec_b64 %[active], vcc\n"
		"ds_read_b128 v[46:49], v53\n"
		"ds_read_b128 v[50:53], v53 offset:16\n"
		"s_xor_b64 exec, exec, %[active]\n"                  // the lanes on cold records
		"s_cbranch_execz 2f\n"
		"s_and_b64 s[82:83], exec, s[80:81]\n"               // cold even lanes
		"s_andn2_b64 s[84:85], exec, s[80:81]\n"             // cold odd lanes
		"s_lshl_b64 s[86:87], s[82:83], 1\n"
		"s_or_b64 s[86:87], s[86:87], s[82:83]\n"            // both lanes of the pairs whose even lane is cold
		"s_lshr_b64 s[88:89], s[84:85], 1\n"
		"s_or_b64 s[88:89], s[88:89], s[84:85]\n"            // both lanes of the pairs whose odd lane is cold
		"s_mov_b64 exec, s[86:87]\n"
		"s_nop 4\n"
		"v_mov_b32_dpp v73, v53 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n"
		"v_add_u32 v73, v73, v72\n"
		"global_load_dwordx4 v[64:67], v73, %[nodes]\n"      // even lane: its record's first half; odd lane: the second
		"s_mov_b64 exec, s[88:89]\n"
		"s_nop 4\n"
		"v_mov_b32_dpp v73, v53 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n"
		"v_add_u32 v73, v73, v72\n"
		"global_load_dwordx4 v[68:71], v73, %[nodes]\n"      // the odd lane's record, the same way
		"s_or_b64 exec, s[86:87], s[88:89]\n"
		"s_not_b64 vcc, s[82:83]\n"                          // vcc = 1: keep
		"s_waitcnt vmcnt(0) lgkmcnt(0)\n"
		// a cold even lane: n0 = its own first load, n1 = what its partner fetched with it
		"v_cndmask_b32 v46, v64, v46, vcc\n"
		"v_cndmask_b32 v47, v65, v47, vcc\n"
		"v_cndmask_b32 v48, v66, v48, vcc\n"
		"v_cndmask_b32 v49, v67, v49, vcc\n"
		"v_cndmask_b32_dpp v50, v64, v50, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		"v_cndmask_b32_dpp v51, v65, v51, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		"v_cndmask_b32_dpp v52, v66, v52, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		"v_cndmask_b32_dpp v53, v67, v53, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"   // the record's last word = the cursor's next value
		"s_not_b64 vcc, s[84:85]\n"
		// a cold odd lane: n0 = what its partner fetched with it, n1 = its own second load
		"v_cndmask_b32_dpp v46, v68, v46, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		"v_cndmask_b32_dpp v47, v69, v47, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		"v_cndmask_b32_dpp v48, v70, v48, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		"v_cndmask_b32_dpp v49, v71, v49, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
		"v_cndmask_b32 v50, v68, v50, vcc\n"
		"v_cndmask_b32 v51, v69, v51, vcc\n"
		"v_cndmask_b32 v52, v70, v52, vcc\n"
		"v_cndmask_b32 v53, v71, v53, vcc\n"
	"2:\n"
		"s_mov_b64 exec, %[active]\n"
		"v_add_u32 %[visits], 1, %[visits]\n"
		"s_waitcnt vmcnt(0) lgkmcnt(0)\n"
		"v_pk_add_f32 v[54:55], v[46:47], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_add_f32 v[56:57], v[48:49], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_add_f32 v[58:59], v[50:51], %[ozz] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_mul_f32 v[54:55], %[ixy], v[54:55]\n"
		"v_pk_mul_f32 v[56:57], %[ixy], v[56:57]\n"
		"v_pk_mul_f32 v[58:59], %[izz], v[58:59]\n"
		"v_min_f32 v60, v54, v56\n"
		"v_min_f32 v61, v55, v57\n"
		"v_min_f32 v62, v58, v59\n"
		"v_max3_f32 v60, v60, v61, v62\n"
		"v_max_f32 v61, v54, v56\n"
		"v_max_f32 v63, v58, v59\n"
		"v_max_f32 v62, v55, v57\n"
		"v_min3_f32 v61, v61, v62, v63\n"
		"v_cmpx_lt_f32 %[eps], v61\n"
		"v_cmpx_gt_f32 %[rayT], v60\n"
		"v_cmpx_le_f32 v60, v61\n"
		"v_cmp_gt_i32 vcc, 0, v52\n"
		"v_cndmask_b32 v53, v52, v53, vcc\n"
		"s_or_b64 %[parkMask], %[parkMask], vcc\n"
		"s_mov_b64 exec, %[active]\n"
		"v_cmp_le_i32 %[mA], 0, v53\n"
		"s_andn2_b64 exec, %[mA], vcc\n"
		"s_bcnt1_i32_b64 %[count], exec\n"
		"s_cmp_gt_i32 %[count], %[keep]\n"
		"s_cbranch_scc1 1b\n"
		"s_mov_b64 exec, %[saved]\n"
		"v_mov_b32 %[ref], v53\n"
		"v_cndmask_b32 %[parked], 0, 1, %[parkMask]\n"
		"v_mov_b32 %[leafWord], v52\n"
		"v_mov_b32 %[leafTNear], v60\n"
		: [ref] "+v"( ref ), [visits] "+v"( visits ), [leafWord] "=v"( leafWord ), [leafTNear] "=v"( leafTNear ), [parked] "=v"( parked ),
		  [saved] "=&s"( saved ), [active] "=&s"( active ), [parkMask] "=&s"( parkMask ), [mA] "=&s"( mA ), [count] "=&s"( count )
		: [oxy] "v"( oxy ), [ozz] "v"( ozz ), [ixy] "v"( ixy ), [izz] "v"( izz ), [rayT] "v"( rayT ), [keep] "s"( keep ),
		  [numHotBytes] "s"( P.numHotBytes ), [nodes] "s"( P.nodes ), [eps] "s"( eps ), [laneHalf] "v"( laneHalf )
		: "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
		  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73",
		  "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "vcc", "scc"
	);
}

#ifdef PBR_HALVES
#define nodePhasePair nodePhaseHalves     // the kernel in phased-lean's place (-DPBR_PAIR_LEAN) calls this one
#endif

// ---- the node phase with TWO walks per lane (round 4, lab: traversal-only probe) ----------------------------
// VERDICT r03 item 2 asks what "two rays per lane" buys: memory-level parallelism without more waves.  This is the
// doubled node phase by itself: every lane carries walk A and walk B (cursors in v53 / v71, records in v[46:53] /
// v[64:71]); an iteration issues A's fetches, then B's, waits once, runs A's slab test and B's.  44 vector + ~22 scalar
// instructions per iteration for up to 128 visits.  Used by diagTraceStreamDual only (lab builds): what it is worth at
// equal occupancy is measured there before any state machine is rebuilt around it.
#ifdef PBR_LAB
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

template<int DUMMY = 0>
PT_DEV void nodePhaseDual(
	const DevParams& P,
	const f2v oxyA, const f2v ozzA, const f2v ixyA, const f2v izzA, float rayTA,
	const f2v oxyB, const f2v ozzB, const f2v ixyB, const f2v izzB, float rayTB,
	int keep, int& refA, int& refB, unsigned& visitsA, unsigned& visitsB,
	int& leafWordA, float& tNearA, int& leafWordB, float& tNearB      // leaf word != 0 on return: the walk parked on that hit leaf
) {
	const float eps = EPSILON5;
	keep = __builtin_amdgcn_readfirstlane( keep );

	asm volatile(
		"s_mov_b64 s[84:85], exec\n"
		"v_cmp_le_i32 s[86:87], 0, %[refA]\n"                // lanes whose walk A goes on
		"v_cmp_le_i32 s[88:89], 0, %[refB]\n"
		"s_mov_b64 s[90:91], 0\n"
		"s_mov_b64 s[92:93], 0\n"
		"v_mov_b32 v53, %[refA]\n"
		"v_mov_b32 v71, %[refB]\n"
	"1:\n"
		"s_mov_b64 exec, s[86:87]\n"
		"s_cbranch_execz 2f\n"
		"v_cmp_gt_i32 vcc, %[numHotBytes], v53\n"
		"s_and_saveexec_b64 s[94:95], vcc\n"
		"ds_read_b128 v[46:49], v53\n"
		"ds_read_b128 v[50:53], v53 offset:16\n"
		"s_xor_b64 exec, exec, s[94:95]\n"
		"global_load_dwordx4 v[46:49], v53, %[nodes]\n"
		"global_load_dwordx4 v[50:53], v53, %[nodes] offset:16\n"
	"2:\n"
		"s_mov_b64 exec, s[88:89]\n"
		"s_cbranch_execz 3f\n"
		"v_cmp_gt_i32 vcc, %[numHotBytes], v71\n"
		"s_and_saveexec_b64 s[94:95], vcc\n"
		"ds_read_b128 v[64:67], v71\n"
		"ds_read_b128 v[68:71], v71 offset:16\n"
		"s_xor_b64 exec, exec, s[94:95]\n"
		"global_load_dwordx4 v[64:67], v71, %[nodes]\n"
		"global_load_dwordx4 v[68:71], v71, %[nodes] offset:16\n"
	"3:\n"
		"s_waitcnt vmcnt(0) lgkmcnt(0)\n"
		// ---- walk A
		"s_mov_b64 exec, s[86:87]\n"
		"s_cbranch_execz 4f\n"
		"v_add_u32 %[visitsA], 1, %[visitsA]\n"
		PT_DUAL_SLAB( "46", "47", "48", "49", "50", "51", "%[oxyA]", "%[ozzA]", "%[ixyA]", "%[izzA]" )
		"v_cmpx_lt_f32 %[eps], v61\n"
		"v_cmpx_gt_f32 %[rayTA], v60\n"
		"v_cmpx_le_f32 v60, v61\n"
		"v_cmp_gt_i32 vcc, 0, v52\n"
		"v_cndmask_b32 v53, v52, v53, vcc\n"
		"s_or_b64 s[90:91], s[90:91], vcc\n"
		"s_mov_b64 exec, vcc\n"                               // the lanes that park now: their leaf word and tNear (v60 is B's next)
		"v_mov_b32 %[leafWordA], v52\n"
		"v_mov_b32 %[tNearA], v60\n"
		"s_mov_b64 exec, s[86:87]\n"
		"v_cmp_le_i32 s[94:95], 0, v53\n"
		"s_andn2_b64 s[86:87], s[94:95], s[90:91]\n"
	"4:\n"
		// ---- walk B
		"s_mov_b64 exec, s[88:89]\n"
		"s_cbranch_execz 5f\n"
		"v_add_u32 %[visitsB], 1, %[visitsB]\n"
		PT_DUAL_SLAB( "64", "65", "66", "67", "68", "69", "%[oxyB]", "%[ozzB]", "%[ixyB]", "%[izzB]" )
		"v_cmpx_lt_f32 %[eps], v61\n"
		"v_cmpx_gt_f32 %[rayTB], v60\n"
		"v_cmpx_le_f32 v60, v61\n"
		"v_cmp_gt_i32 vcc, 0, v70\n"
		"v_cndmask_b32 v71, v70, v71, vcc\n"
		"s_or_b64 s[92:93], s[92:93], vcc\n"
		"s_mov_b64 exec, vcc\n"
		"v_mov_b32 %[leafWordB], v70\n"
		"v_mov_b32 %[tNearB], v60\n"
		"s_mov_b64 exec, s[88:89]\n"
		"v_cmp_le_i32 s[94:95], 0, v71\n"
		"s_andn2_b64 s[88:89], s[94:95], s[92:93]\n"
	"5:\n"
		"s_bcnt1_i32_b64 s96, s[86:87]\n"
		"s_bcnt1_i32_b64 s97, s[88:89]\n"
		"s_add_i32 s96, s96, s97\n"
		"s_cmp_gt_i32 s96, %[keep]\n"
		"s_cbranch_scc1 1b\n"
		"s_mov_b64 exec, s[84:85]\n"
		"v_mov_b32 %[refA], v53\n"
		"v_mov_b32 %[refB], v71\n"
		: [refA] "+v"( refA ), [refB] "+v"( refB ), [visitsA] "+v"( visitsA ), [visitsB] "+v"( visitsB ),
		  [leafWordA] "+v"( leafWordA ), [tNearA] "+v"( tNearA ), [leafWordB] "+v"( leafWordB ), [tNearB] "+v"( tNearB )
		: [oxyA] "v"( oxyA ), [ozzA] "v"( ozzA ), [ixyA] "v"( ixyA ), [izzA] "v"( izzA ), [rayTA] "v"( rayTA ),
		  [oxyB] "v"( oxyB ), [ozzB] "v"( ozzB ), [ixyB] "v"( ixyB ), [izzB] "v"( izzB ), [rayTB] "v"( rayTB ),
		  [keep] "s"( keep ), [numHotBytes] "s"( P.numHotBytes ), [nodes] "s"( P.nodes ), [eps] "s"( eps )
		: "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
		  "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",
		  "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "vcc", "scc"
	);
}

// (nodePhaseDualPipe, the software-pipelined form of nodePhaseDual, became the product's: csrc/pt_dual.hpp.)
#undef PT_DUAL_SLAB
#endif

// ---- the node phase, asynchronous (round 4) --------------------------------------------------------
// nodePhaseAsm ends every iteration in `s_waitcnt vmcnt(0)`: it lasts as long as its SLOWEST lane's fetch.  Measured in
// round 3 on the Sponza-class scene: an L1 miss comes back after 194 cycles on average, an iteration lasts ~900 — with
// ~18 lanes on cold records and an L2 hit rate of 84 %, 96 % of all iterations contain at least one request that goes
// on to the Infinity Cache or HBM, and 62 % of all wave-cycles are spent waiting.  s_waitcnt cannot wait for "most" of
// a wave's lanes; a load to registers has no other way of telling that it has arrived.
//
// A load to LDS has: the bytes are simply there at some point.  Here every lane owns a 32-byte SLOT in LDS (two 16-byte
// halves, PT_SLOT_PLANE apart, lane-linear — the only destination shape an LDS-DMA has: M0 + offset + 16 * lane).  A
// lane whose next record is cold writes a marker into each half of its slot (the record's last word, w1, is a byte
// offset or negative — never 1; the last word of the first half is a box coordinate — never the NaN 0xFFFFFFFF:
// pbr_upload_scene stores every NaN of a box as 0x7FC00000, which no comparison can tell apart), requests the record
// with two `global_load_lds_dwordx4` and goes on polling: every iteration
// reads, for every walking lane, either the staged record (hot) or the lane's slot (cold), and the lanes whose record
// is there — hot, or marker overwritten — take their visit; the others keep their state and are asked again in the
// next iteration.  An iteration starts as soon as P.asyncEighths / 8 of the walking lanes are ready (else the wave
// sleeps 64 cycles and polls again), so a far miss delays ITS lane, not the wave.  No wait on vmcnt anywhere in the loop.
// Nothing orders a ds_read behind a pending LDS-DMA (MI355X_MICROARCH.md, item 7: "a read issued earlier returns the OLD
// LDS bytes, no stall") — which is exactly the behaviour polled for.  Measured with scripts/micro/glds_poll.hip
// (profiles/r04/experiments/glds_poll.txt, 6e9 records per variant, each checked word by word): a lane's 16 bytes land
// at once, but the two halves of a record land IN EITHER ORDER (with a marker in the second half only, 0.02 % of the
// records were consumed torn; with both, none) — hence the two markers.
// Invariant between phases: a walking lane's cursor is either the reference of a staged record (< P.slotBase) or the
// LDS address of its own slot, with the request issued (cold cursors are replaced by the slot address when the request
// goes out; a lane that parks on a leaf has its next record requested before the leaf phase).  Per lane the sequence of
// visits is the reference's; only which iteration a visit falls into changes.
// Registers: v46-v63 as nodePhaseAsm; v64 = address of the second half - 16, v65 = this lane's slot + 12, v66 / v67 = the markers.
// Scalars s84-s99 are the block's own (clobbered): saved exec, walking lanes, ready lanes, parked lanes, a temporary
// mask, counts, the compiler's M0, the error flag (s83).
#if defined( PBR_DBG_NOREARM )
#define PT_ASYNC_REARM ""
#elif defined( PBR_DBG_REARM2 )
#define PT_ASYNC_REARM "ds_write_b32 v65, v67\n" "ds_write_b32 v65, v66 offset:16384\n"
#else
#define PT_ASYNC_REARM "ds_write2st64_b32 v65, v67, v66 offset1:64\n"        // both markers back into this lane's slot (+ 12, + PT_SLOT_PLANE + 12)
#endif
#if defined( PBR_DBG_NOP )
#define PT_ASYNC_NOP "s_nop 7\n"
#else
#define PT_ASYNC_NOP ""
#endif
#define PT_ASYNC_POLL_LIMIT 0x100000

template<int DUMMY = 0>
PT_DEV void nodePhaseAsync(
	const DevParams& P, const f2v oxy, const f2v ozz, const f2v ixy, const f2v izz, float rayT, int keep, int m0a,
	int& ref, unsigned& visits, int& leafWord, float& leafTNear, int& parked, int& err
) {
	const float eps = EPSILON5;
	keep = __builtin_amdgcn_readfirstlane( keep );
	m0a = __builtin_amdgcn_readfirstlane( m0a );         // P.slotBase + 1024 * (wave of the block): + 16 * lane by the hardware

	asm volatile(
		"s_mov_b64 s[84:85], exec\n"
		"s_mov_b64 s[86:87], exec\n"
		"s_mov_b64 s[90:91], 0\n"
		"s_mov_b32 s98, m0\n"
		"s_mov_b32 s97, 0\n"
		"s_mov_b32 s83, 0\n"
		"s_add_u32 s99, %[m0a], 16368\n"                     // second half: M0 + offset:16 + 16 * lane = slot + PT_SLOT_PLANE
		"v_mbcnt_lo_u32_b32 v65, -1, 0\n"
		"v_mbcnt_hi_u32_b32 v65, -1, v65\n"
		"v_lshl_add_u32 v65, v65, 4, %[m0a]\n"
		"v_add_u32 v65, 12, v65\n"                           // this lane's slot + 12: its first marker
		"v_mov_b32 v66, 1\n"
		"v_mov_b32 v67, -1\n"
		"v_add_u32 v64, 16356, v65\n"                        // slot + PT_SLOT_PLANE - 16
		"v_cmp_gt_i32 vcc, %[slotBase], %[ref]\n"            // staged record?  else the cursor is the slot (request issued)
		"v_cndmask_b32 v64, v64, %[ref], vcc\n"
		"s_bcnt1_i32_b64 s95, exec\n"
	"1:\n"
		"s_mul_i32 s96, s95, %[eighths]\n"                    // lanes that must be ready for an iteration to start
		"s_lshr_b32 s96, s96, 3\n"
		"s_max_i32 s96, s96, 1\n"
	"2:\n"
		"ds_read_b128 v[46:49], %[ref]\n"
		"ds_read_b128 v[50:53], v64 offset:16\n"
		"s_waitcnt lgkmcnt(0)\n"
		"v_cmp_ne_u32 vcc, 1, v53\n"                          // the record is there: staged, or BOTH markers are gone
		"v_cmp_ne_u32 s[92:93], -1, v49\n"                   // (the halves land in either order, scripts/micro/glds_poll.hip)
		"s_and_b64 vcc, vcc, s[92:93]\n"
		"s_bcnt1_i32_b64 s94, vcc\n"
		"s_cmp_ge_i32 s94, s96\n"
		"s_cbranch_scc1 3f\n"
		"s_add_i32 s97, s97, 1\n"
		"s_cmp_gt_i32 s97, %[pollLimit]\n"
		"s_cbranch_scc1 8f\n"
		"s_sleep 1\n"
		"s_branch 2b\n"
	"3:\n"
		"s_mov_b64 s[88:89], vcc\n"
		"s_mov_b64 exec, vcc\n"
		PT_ASYNC_REARM
		"v_add_u32 %[visits], 1, %[visits]\n"
		"v_pk_add_f32 v[54:55], v[46:47], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_add_f32 v[56:57], v[48:49], %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_add_f32 v[58:59], v[50:51], %[ozz] neg_lo:[0,1] neg_hi:[0,1]\n"
		"v_pk_mul_f32 v[54:55], %[ixy], v[54:55]\n"
		"v_pk_mul_f32 v[56:57], %[ixy], v[56:57]\n"
		"v_pk_mul_f32 v[58:59], %[izz], v[58:59]\n"
		"v_min_f32 v60, v54, v56\n"
		"v_min_f32 v61, v55, v57\n"
		"v_min_f32 v62, v58, v59\n"
		"v_max3_f32 v60, v60, v61, v62\n"
		"v_max_f32 v61, v54, v56\n"
		"v_max_f32 v63, v58, v59\n"
		"v_max_f32 v62, v55, v57\n"
		"v_min3_f32 v61, v61, v62, v63\n"
		"v_cmpx_lt_f32 %[eps], v61\n"
		"v_cmpx_gt_f32 %[rayT], v60\n"
		"v_cmpx_le_f32 v60, v61\n"
		"v_cmp_le_i32 vcc, 0, v52\n"                          // hit container
		PT_ASYNC_NOP
		"s_andn2_b64 s[92:93], exec, vcc\n"                   // hit leaf: parks
		"s_or_b64 s[90:91], s[90:91], s[92:93]\n"
		"s_mov_b64 exec, s[88:89]\n"
		"v_cndmask_b32 %[ref], v53, v52, vcc\n"               // hit container -> w0, everything else -> w1
		"v_mov_b32 v64, %[ref]\n"
		"v_cmp_gt_i32 vcc, 0, %[ref]\n"                       // the walk has ended
		"s_or_b64 s[92:93], s[92:93], vcc\n"
		"s_andn2_b64 s[86:87], s[86:87], s[92:93]\n"
		"v_cmp_le_i32 vcc, %[slotBase], %[ref]\n"             // next record cold: request it (parked lanes too)
		"s_and_b64 exec, exec, vcc\n"
		"s_cbranch_scc0 4f\n"
		"s_waitcnt lgkmcnt(0)\n"                              // the marker is in the slot before the request leaves
		"s_mov_b32 m0, %[m0a]\n"
		"s_nop 0\n"
		"global_load_lds_dwordx4 %[ref], %[nodes]\n"
		"s_mov_b32 m0, s99\n"
		"s_nop 0\n"
		"global_load_lds_dwordx4 %[ref], %[nodes] offset:16\n"
		"v_add_u32 %[ref], -12, v65\n"
		"v_add_u32 v64, 16356, v65\n"
	"4:\n"
		"s_mov_b64 exec, s[86:87]\n"
		"s_bcnt1_i32_b64 s95, s[86:87]\n"
		"s_cmp_gt_i32 s95, %[keep]\n"
		"s_cbranch_scc1 1b\n"
	"5:\n"
		"s_mov_b64 exec, s[84:85]\n"
		"s_mov_b32 m0, s98\n"
		"v_cndmask_b32 %[parked], 0, 1, s[90:91]\n"
		"v_mov_b32 %[leafWord], v52\n"
		"v_mov_b32 %[leafTNear], v60\n"
		"v_mov_b32 %[err], s83\n"
		"s_branch 7f\n"
	"8:\n"                                                    // the poll limit: never hang a GPU
		"s_waitcnt vmcnt(0)\n"
		"s_cmp_eq_u32 s83, 0\n"
		"s_mov_b32 s83, 1\n"
		"s_mov_b32 s97, 0\n"
		"s_cbranch_scc1 2b\n"
		"v_mov_b32 %[ref], -1\n"
		"s_branch 5b\n"
	"7:\n"
		: [ref] "+v"( ref ), [visits] "+v"( visits ), [leafWord] "=v"( leafWord ), [leafTNear] "=v"( leafTNear ), [parked] "=v"( parked ), [err] "=v"( err )
		: [oxy] "v"( oxy ), [ozz] "v"( ozz ), [ixy] "v"( ixy ), [izz] "v"( izz ), [rayT] "v"( rayT ), [keep] "s"( keep ),
		  [slotBase] "s"( P.slotBase ), [nodes] "s"( P.nodes ), [eps] "s"( eps ), [m0a] "s"( m0a ), [eighths] "s"( P.asyncEighths ),
		  [pollLimit] "n"( PT_ASYNC_POLL_LIMIT )
		: "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
		  "v64", "v65", "v66", "v67", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
		  "vcc", "scc", "memory"
	);
}
