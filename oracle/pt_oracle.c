/*
 * pt_oracle.c — CPU restatement of the reference path-tracing kernel.
 *
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h).  PARITY UNPINNED (see pt_oracle.h).
 *
 * Follows /root/reference/source/opencl/{pathtracing,pt_utils,pt_rgb,pt_brdf,
 * pt_intersect,pt_bvh}.cl one function at a time; every function cites the
 * file:line it restates, pt_phongtess.cl (PHONGTESS=1, off in the reference's
 * config.json:105) included.
 *
 * ONE section is NOT a restatement of the reference: "Ray-ordered walk" (cfg.traversal != 0) states the product's
 * opt-in walk order over the same flat tree, so that the HIP path of that mode has a bit-exact checker; it is marked
 * where it stands, the default (cfg.traversal == 0) never enters it, and what it owes the reference's own order is tested
 * separately (tests/test_walk_order_cpu.py).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -mfma -fopenmp -shared -fPIC
 */
#include "pt_oracle.h"

#include <math.h>
#include <string.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* Deterministic math layer                                                  */
/* ------------------------------------------------------------------------- */
/*
 * One bit-exact definition for every OpenCL builtin the kernels use whose
 * result is implementation-defined.  Only IEEE +,-,*,/,sqrt, fmaf/fma, floor,
 * rint and comparisons are used, so gcc/x86-64 and hipcc/gfx950 agree bit for
 * bit when both are built with contraction off.
 *
 *   native_recip(x)      = 1.0f / x                 (IEEE)
 *   native_divide(a,b)   = a / b                    (IEEE)
 *   native_sqrt(x)       = sqrtf(x)                 (IEEE)
 *   dot(a,b)             = fma(az,bz, fma(ay,by, ax*bx))
 *   cross(a,b).x         = fma(ay,bz, -(az*by))     (cyclic)
 *   fast_normalize(v)    = v * (1.0f / sqrtf(dot(v,v)))
 *   length(v)            = sqrtf(dot(v,v))
 *   fma(a,b,c)           = fmaf per component       (where the reference writes fma)
 *   fract(x)             = fmin(x - floor(x), 0x1.fffffep-1f)   (OpenCL 1.1 6.11.2)
 *   mix(x,y,a)           = x + (y - x) * a                       (OpenCL 1.1 6.11.4)
 *   clamp(x,lo,hi)       = fmin(fmax(x,lo),hi)                   (OpenCL 1.1 6.11.4)
 *   max(x,y)             = x < y ? y : x                         (OpenCL 1.1 6.11.4)
 *   native_sin/cos       = det_sincos below (3-term Cody-Waite + degree-7/8 minimax)
 *   native_tan           = det_sin / det_cos
 *   acos, atan           = single-precision minimax forms below (<= 4 / <= 5 ulp)
 *   pow                  = exp2(y * log2(x)) evaluated in binary64 (<= 1 ulp)
 *   cbrt                 = sign(x) * exp2(log2(|x|) / 3) evaluated in binary64 (<= 1 ulp)
 * Double-typed literals in the reference (M_PI, M_PI_2, M_1_PI: the kernels were
 * written for a device with cl_khr_fp64) promote the surrounding expression to
 * binary64 exactly as C does; the result is rounded once where the reference
 * stores it in a float.
 */

#define ORC_INF (__builtin_inff())

static const double M_PI_D = 0x1.921fb54442d18p+1;
static const double M_PI_2_D = 0x1.921fb54442d18p+0;
static const double M_1_PI_D = 0x1.45f306dc9c883p-2;

#define EPSILON5 0.00001f
#define NI_AIR 1.00028f
#define PI_X2 6.28318530718f

static inline float det_rcp( float x ) { return 1.0f / x; }
static inline float det_div( float a, float b ) { return a / b; }
static inline float det_sqrt( float x ) { return sqrtf( x ); }
static inline float det_max( float x, float y ) { return ( x < y ) ? y : x; }
static inline float det_clamp( float x, float lo, float hi ) { return fminf( fmaxf( x, lo ), hi ); }

/* sin and cos of x.  k = rint(x*2/pi); r = x - k*pi/2 in three fma steps;
 * polynomials on [-pi/4, pi/4]; quadrant from k mod 4 (computed in float so it
 * is defined for every finite k).  |x| > 1e8 or non-finite: x is replaced by
 * x*0 (=> +-0 or NaN). */
static void det_sincos( float x, float* sn, float* cs ) {
	if( !( fabsf( x ) <= 1.0e8f ) ) {
		x = x * 0.0f;
	}

	const float k = rintf( x * 0x1.45f306p-1f );
	float r = fmaf( -k, 0x1.921fb6p+0f, x );
	r = fmaf( -k, -0x1.777a5cp-25f, r );
	r = fmaf( -k, -0x1.ee59dap-50f, r );
	const float z = r * r;

	float ps = fmaf( -1.9515295891e-4f, z, 8.3321608736e-3f );
	ps = fmaf( ps, z, -1.6666654611e-1f );
	const float s = fmaf( ps * z, r, r );

	float pc = fmaf( 2.443315711809948e-5f, z, -1.388731625493765e-3f );
	pc = fmaf( pc, z, 4.166664568298827e-2f );
	const float c = fmaf( pc * z, z, fmaf( -0.5f, z, 1.0f ) );

	const int q = (int) ( k - 4.0f * floorf( k * 0.25f ) );

	*sn = ( q == 0 ) ? s : ( q == 1 ) ? c : ( q == 2 ) ? -s : -c;
	*cs = ( q == 0 ) ? c : ( q == 1 ) ? -s : ( q == 2 ) ? -c : s;
}

static inline float det_sin( float x ) { float s, c; det_sincos( x, &s, &c ); return s; }
static inline float det_cos( float x ) { float s, c; det_sincos( x, &s, &c ); return c; }
static inline float det_tan( float x ) { float s, c; det_sincos( x, &s, &c ); return s / c; }

/* asin on |x| <= 0.5 */
static inline float det_asin_core( float x ) {
	const float z = x * x;
	float p = fmaf( 4.2163199048e-2f, z, 2.4181311049e-2f );
	p = fmaf( p, z, 4.5470025998e-2f );
	p = fmaf( p, z, 7.4953002686e-2f );
	p = fmaf( p, z, 1.6666752422e-1f );
	return fmaf( p * z, x, x );
}

static float det_acos( float x ) {
	if( x < -0.5f ) {
		return 0x1.921fb6p+1f - 2.0f * det_asin_core( sqrtf( 0.5f * ( 1.0f + x ) ) );
	}
	if( x > 0.5f ) {
		return 2.0f * det_asin_core( sqrtf( 0.5f * ( 1.0f - x ) ) );
	}
	/* also the NaN path: NaN compares false twice and propagates through the core */
	return ( 0x1.921fb6p+0f - det_asin_core( x ) ) + -0x1.777a5cp-25f;
}

static float det_atan( float xx ) {
	const float ax = fabsf( xx );
	float x, y0;

	if( ax > 2.414213562373095f ) {
		y0 = 0x1.921fb6p+0f;
		x = -( 1.0f / ax );
	}
	else if( ax > 0.4142135623730950f ) {
		y0 = 0x1.921fb6p-1f;
		x = ( ax - 1.0f ) / ( ax + 1.0f );
	}
	else {
		y0 = 0.0f;
		x = ax;
	}

	const float z = x * x;
	float p = fmaf( 8.05374449538e-2f, z, -1.38776856032e-1f );
	p = fmaf( p, z, 1.99777106478e-1f );
	p = fmaf( p, z, -3.33329491539e-1f );
	const float y = y0 + fmaf( p * z, x, x );

	return copysignf( y, xx );
}

static inline uint64_t d2u( double d ) { uint64_t u; memcpy( &u, &d, 8 ); return u; }
static inline double u2d( uint64_t u ) { double d; memcpy( &d, &u, 8 ); return d; }

/* log2 of a positive, finite, normal binary64 */
static double det_log2_d( double a ) {
	const uint64_t bits = d2u( a );
	int e = (int) ( ( bits >> 52 ) & 0x7ff ) - 1023;
	double m = u2d( ( bits & 0x000fffffffffffffULL ) | 0x3ff0000000000000ULL );

	if( m > 0x1.6a09e667f3bcdp+0 ) {
		m *= 0.5;
		e += 1;
	}

	const double s = ( m - 1.0 ) / ( m + 1.0 );
	const double s2 = s * s;
	double p = 0x1.e1e1e1e1e1e1ep-5;              /* 1/17 */
	p = fma( p, s2, 0x1.1111111111111p-4 );       /* 1/15 */
	p = fma( p, s2, 0x1.3b13b13b13b14p-4 );       /* 1/13 */
	p = fma( p, s2, 0x1.745d1745d1746p-4 );       /* 1/11 */
	p = fma( p, s2, 0x1.c71c71c71c71cp-4 );       /* 1/9 */
	p = fma( p, s2, 0x1.2492492492492p-3 );       /* 1/7 */
	p = fma( p, s2, 0x1.999999999999ap-3 );       /* 1/5 */
	p = fma( p, s2, 0x1.5555555555555p-2 );       /* 1/3 */
	p = fma( p, s2, 1.0 );
	const double ln_m = 2.0 * s * p;

	return fma( ln_m, 0x1.71547652b82fep+0, (double) e );
}

/* 2^t for t in [-160, 130] */
static double det_exp2_d( double t ) {
	const double n = rint( t );
	const double g = ( t - n ) * 0x1.62e42fefa39efp-1;
	double p = 0x1.6124613a86d09p-33;             /* 1/13! */
	p = fma( p, g, 0x1.1eed8eff8d898p-29 );
	p = fma( p, g, 0x1.ae64567f544e4p-26 );
	p = fma( p, g, 0x1.27e4fb7789f5cp-22 );
	p = fma( p, g, 0x1.71de3a556c734p-19 );
	p = fma( p, g, 0x1.a01a01a01a01ap-16 );
	p = fma( p, g, 0x1.a01a01a01a01ap-13 );
	p = fma( p, g, 0x1.6c16c16c16c17p-10 );
	p = fma( p, g, 0x1.1111111111111p-7 );
	p = fma( p, g, 0x1.5555555555555p-5 );
	p = fma( p, g, 0x1.5555555555555p-3 );
	p = fma( p, g, 0.5 );
	p = fma( p, g, 1.0 );
	p = fma( p, g, 1.0 );
	const double scale = u2d( (uint64_t) ( (int64_t) n + 1023 ) << 52 );

	return p * scale;
}

/* pow with C99 / OpenCL special cases */
static float det_pow( float x, float y ) {
	if( y == 0.0f || x == 1.0f ) {
		return 1.0f;
	}
	if( x != x || y != y ) {
		return x + y;
	}

	const float ay = fabsf( y );
	const int y_is_int = ( ay >= 0x1p24f ) || ( floorf( ay ) == ay );
	/* odd integer: only possible below 2^24 */
	const int y_is_odd = y_is_int && ( ay < 0x1p24f ) && ( fmodf( ay, 2.0f ) == 1.0f );
	const float ax = fabsf( x );
	float sign = 1.0f;

	if( x < 0.0f || ( x == 0.0f && signbit( x ) ) ) {
		if( y_is_odd ) {
			sign = -1.0f;
		}
		else if( !y_is_int && ax != 0.0f && ax != ORC_INF ) {
			return ORC_INF - ORC_INF; /* NaN */
		}
	}

	if( ax == 1.0f ) {
		return sign;
	}

	double l;

	if( ax == 0.0f ) {
		l = -(double) ORC_INF;
	}
	else if( ax == ORC_INF ) {
		l = (double) ORC_INF;
	}
	else {
		l = det_log2_d( (double) ax );
	}

	double t = (double) y * l;
	t = ( t > 130.0 ) ? 130.0 : t;
	t = ( t < -160.0 ) ? -160.0 : t;

	return sign * (float) det_exp2_d( t );
}

/* cbrt (solveCubic, pt_utils.cl:153): through the binary64 log2 / exp2 of det_pow */
static float det_cbrt( float x ) {
	const float ax = fabsf( x );

	if( x != x || ax == 0.0f || ax == ORC_INF ) {
		return x;
	}

	return copysignf( (float) det_exp2_d( det_log2_d( (double) ax ) / 3.0 ), x );
}

static inline float det_fract( float x ) {
	return fminf( x - floorf( x ), 0x1.fffffep-1f );
}


/* ------------------------------------------------------------------------- */
/* float3 helpers                                                            */
/* ------------------------------------------------------------------------- */

typedef struct { float x, y, z; } v3;

static inline v3 V3( float x, float y, float z ) { v3 r = { x, y, z }; return r; }
static inline v3 v3_from4( orc_float4 a ) { return V3( a.x, a.y, a.z ); }
static inline v3 v3_add( v3 a, v3 b ) { return V3( a.x + b.x, a.y + b.y, a.z + b.z ); }
static inline v3 v3_sub( v3 a, v3 b ) { return V3( a.x - b.x, a.y - b.y, a.z - b.z ); }
static inline v3 v3_mul( v3 a, v3 b ) { return V3( a.x * b.x, a.y * b.y, a.z * b.z ); }
static inline v3 v3_scale( v3 a, float s ) { return V3( a.x * s, a.y * s, a.z * s ); }
static inline v3 v3_neg( v3 a ) { return V3( -a.x, -a.y, -a.z ); }
static inline v3 v3_yzx( v3 a ) { return V3( a.y, a.z, a.x ); }

static inline float v3_dot( v3 a, v3 b ) {
	return fmaf( a.z, b.z, fmaf( a.y, b.y, a.x * b.x ) );
}

static inline v3 v3_cross( v3 a, v3 b ) {
	return V3(
		fmaf( a.y, b.z, -( a.z * b.y ) ),
		fmaf( a.z, b.x, -( a.x * b.z ) ),
		fmaf( a.x, b.y, -( a.y * b.x ) )
	);
}

static inline v3 v3_normalize( v3 a ) {
	const float inv = 1.0f / sqrtf( v3_dot( a, a ) );
	return v3_scale( a, inv );
}

/* fma( scalar, vec, vec ) as the reference writes it */
static inline v3 v3_fma_s( float s, v3 a, v3 b ) {
	return V3( fmaf( s, a.x, b.x ), fmaf( s, a.y, b.y ), fmaf( s, a.z, b.z ) );
}

/* reflect macro, pt_utils.cl:426: dir - 2.0f * dot( normal, dir ) * normal */
static inline v3 reflect3( v3 dir, v3 normal ) {
	const float s = 2.0f * v3_dot( normal, dir );
	return v3_sub( dir, v3_scale( normal, s ) );
}


/* ------------------------------------------------------------------------- */
/* Kernel state                                                              */
/* ------------------------------------------------------------------------- */

/* ray4, pt_header.cl:24-30 */
typedef struct {
	v3 origin, dir, normal;
	float t;
	int hitFace;
} ray4;

/* Material in one shape for both BRDFs (pt_header.cl:79-109).
 * BRDF 0: d, Ni, p, rough.   BRDF 1: d, Ni, nu, nv, Rs, Rd. */
typedef struct {
	float d, Ni;
	float p, rough;        /* Schlick */
	float nu, nv, Rs, Rd;  /* Shirley-Ashikhmin */
	v3 rgbDiff, rgbSpec;
} mtl_t;

typedef struct {
	const orc_scene* scene;
	const orc_config* cfg;
	float dbg_faces, dbg_nodes;  /* scene->debugColor.x / .y, pt_header.cl:75 */
	uint64_t n_hits;
} ctx_t;

static mtl_t load_material( const ctx_t* c, uint32_t index ) {
	mtl_t m;
	memset( &m, 0, sizeof( m ) );

	if( c->cfg->brdf == 0 ) {
		const orc_material_schlick* s = (const orc_material_schlick*) c->scene->materials + index;
		m.d = s->data[0]; m.Ni = s->data[1]; m.p = s->data[2]; m.rough = s->data[3];
		m.rgbDiff = v3_from4( s->rgbDiff ); m.rgbSpec = v3_from4( s->rgbSpec );
	}
	else {
		const orc_material_sa* s = (const orc_material_sa*) c->scene->materials + index;
		m.d = s->data[0]; m.Ni = s->data[1]; m.nu = s->data[2]; m.nv = s->data[3];
		m.Rs = s->data[4]; m.Rd = s->data[5];
		m.rgbDiff = v3_from4( s->rgbDiff ); m.rgbSpec = v3_from4( s->rgbSpec );
	}

	return m;
}


/* ------------------------------------------------------------------------- */
/* pt_utils.cl                                                               */
/* ------------------------------------------------------------------------- */

/* rand, pt_utils.cl:39-44 */
static inline float rnd( float* seed ) {
	*seed += 1.0f;
	return det_fract( det_sin( *seed ) * 43758.5453123f );
}

/* fresnel, pt_utils.cl:53-56:  c + ( 1 - c ) * v * v * v * v * v */
static inline float fresnel( float u, float c ) {
	const float v = 1.0f - u;
	return c + ( 1.0f - c ) * v * v * v * v * v;
}

/* extendDepth, pt_utils.cl:89-96 */
static inline int extendDepth( const ctx_t* c, const mtl_t* mtl, float* seed ) {
	if( c->cfg->brdf == 1 ) {
		return ( fmaxf( mtl->nu, mtl->nv ) >= 50.0f );
	}
	return ( mtl->rough < rnd( seed ) );
}

/* jitter, pt_utils.cl:306-318 */
static v3 jitter( v3 nl, float phi, float sina, float cosa ) {
	const v3 u = v3_normalize( v3_cross( v3_yzx( nl ), nl ) );
	const v3 v = v3_normalize( v3_cross( nl, u ) );
	float sp, cp;
	det_sincos( phi, &sp, &cp );

	const v3 w = v3_normalize( v3_add( v3_scale( u, cp ), v3_scale( v, sp ) ) );

	return v3_normalize( v3_add( v3_scale( w, sina ), v3_scale( nl, cosa ) ) );
}

/* antiAliasing, pt_utils.cl:327-337 */
static void antiAliasing( const ctx_t* c, ray4* ray, float pxDim, float* seed ) {
	const float r = rnd( seed );
	const float phi = PI_X2 * rnd( seed );
	const v3 aaDir = jitter( ray->dir, phi, det_sqrt( r ), det_sqrt( 1.0f - r ) );

	/* ray->dir + aaDir * pxDim * ANTI_ALIASING */
	const v3 off = v3_scale( v3_scale( aaDir, pxDim ), c->cfg->anti_aliasing );
	ray->dir = v3_normalize( v3_add( ray->dir, off ) );
}

/* depthOfField, pt_utils.cl:348-373 */
static void depthOfField( ray4* ray, const orc_camera* cam, float tObject, float tFocus, float* seed ) {
	if( tObject == ORC_INF ) {
		tObject = 1000.0f;
	}
	if( tFocus == ORC_INF ) {
		tFocus = 1000.0f;
	}

	if( tObject > 0.0f ) {
		const float aperture = cam->lense[0] / cam->lense[1];
		const float radius = rnd( seed ) * aperture * 0.5f;
		const float angle = PI_X2 * rnd( seed );
		float sa, ca;
		det_sincos( angle, &sa, &ca );
		const float x = radius * ca;
		const float y = radius * sa;

		/* ray->origin + x * cam->u + y * cam->v */
		ray->origin = v3_add(
			v3_add( ray->origin, v3_scale( v3_from4( cam->u ), x ) ),
			v3_scale( v3_from4( cam->v ), y )
		);

		const v3 hitFocalPlane = v3_fma_s( tFocus, ray->dir, v3_from4( cam->eye ) );
		ray->dir = v3_normalize( v3_sub( hitFocalPlane, ray->origin ) );
	}
}

/* russianRoulette, pt_utils.cl:385-387 — && short-circuits: rand only if first clause holds */
static inline int russianRoulette( int depth, int depthAdded, float maxValColor, float* seed ) {
	return ( depth > 2 + depthAdded && maxValColor < rnd( seed ) );
}

/* refract, pt_utils.cl:436-465 */
static v3 refract3( const ray4* ray, const mtl_t* mtl, float* seed ) {
	const int into = ( v3_dot( ray->normal, v3_neg( ray->dir ) ) > 0.0f );
	const v3 nl = into ? ray->normal : v3_neg( ray->normal );

	const float m1 = into ? NI_AIR : mtl->Ni;
	const float m2 = into ? mtl->Ni : NI_AIR;
	const float m = det_div( m1, m2 );

	const float cosI = -v3_dot( nl, ray->dir );
	const float sinT2 = m * m * ( 1.0f - cosI * cosI );

	if( sinT2 >= 1.0f ) {
		return reflect3( ray->dir, nl );
	}

	const float sqrtCosT = det_sqrt( 1.0f - sinT2 );
	const float r0 = det_div( m1 - m2, m1 + m2 );
	const float c = ( m1 > m2 ) ? sqrtCosT : cosI;
	const float reflectance = fresnel( c, r0 * r0 );

	if( reflectance < rnd( seed ) ) {
		/* m * ray->dir + ( m * cosI - sqrtCosT ) * nl */
		const float k = m * cosI - sqrtCosT;
		return v3_add( v3_scale( ray->dir, m ), v3_scale( nl, k ) );
	}

	return reflect3( ray->dir, nl );
}


/* ------------------------------------------------------------------------- */
/* pt_brdf.cl — BRDF 0: Schlick                                              */
/* ------------------------------------------------------------------------- */

/* Z, pt_brdf.cl:11-14 */
static inline float sch_Z( float t, float r ) {
	const float x = 1.0f + r * t * t - t * t;
	return ( x == 0.0f ) ? 0.0f : det_div( r, x * x );
}

/* A, pt_brdf.cl:23-28 */
static inline float sch_A( float w, float p ) {
	const float p2 = p * p;
	const float w2 = w * w;
	const float x = p2 - p2 * w2 + w2;
	return ( x == 0.0f ) ? 0.0f : det_sqrt( det_div( p, x ) );
}

/* G, pt_brdf.cl:37-40 */
static inline float sch_G( float v, float r ) {
	const float x = r - r * v + v;
	return ( x == 0.0f ) ? 0.0f : det_div( v, x );
}

/* B2, pt_brdf.cl:71-80 */
static inline float sch_B2( float t, float vOut, float vIn, float w, float r, float p ) {
	const float gp = sch_G( vOut, r ) * sch_G( vIn, r );
	const float obstructed = gp * sch_Z( t, r ) * sch_A( w, p );
	const float reemission = 1.0f - gp;
	return obstructed + reemission;
}

/* D, pt_brdf.cl:93-112.  M_PI / M_1_PI are binary64 literals. */
static float sch_D( float t, float vOut, float vIn, float w, float r, float p ) {
	const float b = 4.0f * r * ( 1.0f - r );
	const float a = ( r < 0.5f ) ? 0.0f : 1.0f - b;
	const float c = ( r < 0.5f ) ? 1.0f - b : 0.0f;

	const float d = (float) ( (double) 4.0f * M_PI_D * (double) vOut * (double) vIn );

	const float lam = (float) ( (double) a * M_1_PI_D );
	const float ani = ( b == 0.0f || d == 0.0f )
	                ? 0.0f
	                : det_div( b, d ) * sch_B2( t, vOut, vIn, w, r, p );
	const float fres = ( vIn == 0.0f ) ? 0.0f : det_div( c, vIn );

	return lam + ani + fres;
}

/* brdfSchlick, pt_brdf.cl:125-150 */
static float brdfSchlick(
	const mtl_t* mtl, const ray4* rayLightOut, const ray4* rayLightIn,
	const v3* normal, float* u, float* pdf
) {
	const v3 V_IN = rayLightIn->dir;
	const v3 V_OUT = v3_neg( rayLightOut->dir );

	const v3 un = v3_normalize( v3_cross( v3_yzx( *normal ), *normal ) );

	const v3 h = v3_normalize( v3_add( V_OUT, V_IN ) );
	const float t = v3_dot( h, *normal );
	const float vIn = v3_dot( V_IN, *normal );
	const float vOut = v3_dot( V_OUT, *normal );
	const v3 hp = v3_normalize( v3_cross( v3_cross( h, *normal ), *normal ) );
	const float w = v3_dot( un, hp );

	*u = v3_dot( h, V_OUT );
	/* native_divide( t, 4.0f * M_PI * dot( V_OUT, h ) ): binary64 product rounded to float */
	*pdf = det_div( t, (float) ( (double) 4.0f * M_PI_D * (double) v3_dot( V_OUT, h ) ) );

	return sch_D( t, vOut, vIn, w, mtl->rough, mtl->p );
}

/* newRaySchlick, pt_brdf.cl:160-208 */
static v3 newRaySchlick( const ray4* ray, const mtl_t* mtl, float* seed ) {
	if( mtl->rough == 0.0f ) {
		return reflect3( ray->dir, ray->normal );
	}

	float a = rnd( seed );
	float b = rnd( seed );
	const float iso2 = mtl->p * mtl->p;
	const float alpha = det_acos( det_sqrt( det_div( a, mtl->rough - a * mtl->rough + a ) ) );
	float phi;

	/* phi = M_PI_2 * native_sqrt( ... ): binary64 product, rounded on assignment */
	#define SCH_PHI( bb ) (float) ( M_PI_2_D * (double) det_sqrt( det_div( iso2 * ( bb ), 1.0f - ( bb ) + ( bb ) * iso2 ) ) )

	if( b < 0.25f ) {
		b = 1.0f - 4.0f * ( 0.25f - b );
		const float b2 = b * b;
		phi = SCH_PHI( b2 );
	}
	else if( b < 0.5f ) {
		b = 1.0f - 4.0f * ( 0.5f - b );
		const float b2 = b * b;
		phi = SCH_PHI( b2 );
		phi = (float) ( M_PI_D - (double) phi );
	}
	else if( b < 0.75f ) {
		b = 1.0f - 4.0f * ( 0.75f - b );
		const float b2 = b * b;
		phi = SCH_PHI( b2 );
		phi = (float) ( M_PI_D + (double) phi );
	}
	else {
		b = 1.0f - 4.0f * ( 1.0f - b );
		const float b2 = b * b;
		phi = SCH_PHI( b2 );
		phi = (float) ( (double) 2.0f * M_PI_D - (double) phi );
	}

	#undef SCH_PHI

	if( mtl->p < 1.0f ) {
		phi = (float) ( (double) phi + M_PI_2_D );
	}

	float sa, ca;
	det_sincos( alpha, &sa, &ca );
	const v3 H = jitter( ray->normal, phi, sa, ca );
	v3 newRay = reflect3( ray->dir, H );

	if( v3_dot( newRay, ray->normal ) <= 0.0f ) {
		newRay = jitter( ray->normal, PI_X2 * rnd( seed ), det_sqrt( a ), det_sqrt( 1.0f - a ) );
	}

	return newRay;
}


/* ------------------------------------------------------------------------- */
/* pt_brdf.cl — BRDF 1: Shirley-Ashikhmin                                    */
/* ------------------------------------------------------------------------- */

/* brdfShirleyAshikhmin, pt_brdf.cl:228-268 */
static void brdfShirleyAshikhmin(
	float nu, float nv, float Rs, float Rd,
	const ray4* rayLightOut, const ray4* rayLightIn, const v3* normal,
	float* brdfSpec, float* brdfDiff, float* dotHK1, float* pdf
) {
	(void) Rs;
	const v3 un = v3_normalize( v3_cross( v3_yzx( *normal ), *normal ) );
	const v3 vn = v3_normalize( v3_cross( *normal, un ) );

	const v3 k1 = rayLightIn->dir;
	const v3 k2 = v3_neg( rayLightOut->dir );
	const v3 h = v3_normalize( v3_add( k1, k2 ) );

	const float dotHU = v3_dot( h, un );
	const float dotHV = v3_dot( h, vn );
	const float dotHN = v3_dot( h, *normal );
	const float dotNK1 = v3_dot( *normal, k1 );
	const float dotNK2 = v3_dot( *normal, k2 );
	*dotHK1 = v3_dot( h, k1 );

	float ps_e = nu * dotHU * dotHU + nv * dotHV * dotHV;
	ps_e = ( dotHN == 1.0f ) ? 0.0f : det_div( ps_e, 1.0f - dotHN * dotHN );
	/* native_sqrt(..) * 0.125f * M_1_PI: the last product is binary64 */
	const float ps0 = (float) ( (double) ( det_sqrt( ( nu + 1.0f ) * ( nv + 1.0f ) ) * 0.125f ) * M_1_PI_D );
	const float ps1_num = det_pow( dotHN, ps_e );
	const float ps1 = det_div( ps1_num, ( *dotHK1 ) * fmaxf( dotNK1, dotNK2 ) );

	float pd = Rd * 0.38750768752f;
	const float a = 1.0f - dotNK1 * 0.5f;
	const float b = 1.0f - dotNK2 * 0.5f;
	pd *= 1.0f - a * a * a * a * a;
	pd *= 1.0f - b * b * b * b * b;

	*brdfSpec = ps0 * ps1;
	*brdfDiff = pd;

	const float ph = ps0 * ps1_num;
	*pdf = det_div( ph, *dotHK1 );
}

/* newRayShirleyAshikhmin, pt_brdf.cl:278-330 */
static v3 newRayShirleyAshikhmin( const ray4* ray, const mtl_t* mtl, float* seed ) {
	float a = rnd( seed );
	const float b = rnd( seed );
	float phi_flip = (float) M_PI_D;
	float phi_flipf = 1.0f;
	float aMax = 1.0f;

	if( a < 0.25f ) {
		aMax = 0.25f;
		phi_flip = 0.0f;
	}
	else if( a < 0.5f ) {
		aMax = 0.5f;
		phi_flipf = -1.0f;
	}
	else if( a < 0.75f ) {
		aMax = 0.75f;
	}
	else {
		phi_flip = (float) ( (double) 2.0f * M_PI_D );
		phi_flipf = -1.0f;
	}

	a = 1.0f - 4.0f * ( aMax - a );

	const float phi = det_atan(
		det_sqrt( det_div( mtl->nu + 1.0f, mtl->nv + 1.0f ) ) *
		det_tan( (float) ( M_PI_2_D * (double) a ) )
	);
	const float phi_full = phi_flip + phi_flipf * phi;

	float sinphi, cosphi;
	det_sincos( phi, &sinphi, &cosphi );
	const float theta_e = det_rcp( mtl->nu * cosphi * cosphi + mtl->nv * sinphi * sinphi + 1.0f );
	const float theta = det_acos( det_pow( 1.0f - b, theta_e ) );

	const v3 normal = ( mtl->d < 1.0f || v3_dot( ray->normal, v3_neg( ray->dir ) ) >= 0.0f )
	                ? ray->normal : v3_neg( ray->normal );

	float st, ct;
	det_sincos( theta, &st, &ct );
	const v3 h = jitter( normal, phi_full, st, ct );
	const v3 spec = reflect3( ray->dir, h );
	const v3 diff = jitter( normal, PI_X2 * rnd( seed ), det_sqrt( b ), det_sqrt( 1.0f - b ) );

	return ( v3_dot( spec, normal ) <= 0.0f ) ? diff : spec;
}

/* getNewRay, pt_brdf.cl:344-378.  The reference leaves newRay.hitFace and
 * newRay.normal uninitialised (UB when hitFace is read after a miss,
 * pathtracing.cl:264); here they are defined as 0 / (0,0,0). */
static ray4 getNewRay( const ctx_t* c, const ray4* ray, const mtl_t* mtl, float* seed, int* addDepth ) {
	ray4 newRay;
	newRay.t = ORC_INF;
	newRay.hitFace = 0;
	newRay.normal = V3( 0.0f, 0.0f, 0.0f );
	newRay.origin = v3_fma_s( ray->t, ray->dir, ray->origin );

	/* && short-circuits: rand only if d < 1 */
	const int doTransRefr = ( mtl->d < 1.0f && mtl->d <= rnd( seed ) );

	*addDepth = ( *addDepth || doTransRefr );

	if( doTransRefr ) {
		newRay.dir = refract3( ray, mtl, seed );
	}
	else if( c->cfg->brdf == 0 ) {
		newRay.dir = newRaySchlick( ray, mtl, seed );
	}
	else {
		newRay.dir = newRayShirleyAshikhmin( ray, mtl, seed );
	}

	return newRay;
}


/* ------------------------------------------------------------------------- */
/* pt_intersect.cl                                                           */
/* ------------------------------------------------------------------------- */

/* intersectBox, pt_intersect.cl:11-25 */
static inline int intersectBox(
	const ray4* ray, const v3* invDir, orc_float4 bbMin, orc_float4 bbMax,
	float* tNear, float* tFar
) {
	const v3 t1 = v3_mul( v3_sub( v3_from4( bbMin ), ray->origin ), *invDir );
	v3 tMax = v3_mul( v3_sub( v3_from4( bbMax ), ray->origin ), *invDir );
	const v3 tMin = V3( fminf( t1.x, tMax.x ), fminf( t1.y, tMax.y ), fminf( t1.z, tMax.z ) );
	tMax = V3( fmaxf( t1.x, tMax.x ), fmaxf( t1.y, tMax.y ), fmaxf( t1.z, tMax.z ) );

	*tNear = fmaxf( fmaxf( tMin.x, tMin.y ), tMin.z );
	*tFar = fminf( fminf( tMax.x, tMax.y ), fminf( tMax.z, *tFar ) );

	return ( *tNear <= *tFar );
}

/* intersectSphere, pt_intersect.cl:37-77.  Compares d2 with r (not r*r) as the reference does. */
static int intersectSphere( const ray4* ray, v3 pos, float r, float* tNear, float* tFar ) {
	const v3 L = v3_sub( pos, ray->origin );
	const float tca = v3_dot( L, ray->dir );

	if( tca < 0.0f ) {
		return 0;
	}

	const float d2 = v3_dot( L, L ) - tca * tca;

	if( d2 > r ) {
		return 0;
	}

	const float thc = det_sqrt( r - d2 );
	float t0 = tca - thc;
	float t1 = tca + thc;

	if( t0 > t1 ) {
		const float tmp = t0; t0 = t1; t1 = tmp;
	}

	if( t0 < 0.0f ) {
		t0 = t1;

		if( t0 < 0.0f ) {
			return 0;
		}
	}

	*tNear = t0;
	*tFar = t1;

	return 1;
}

/* flatTriAndRayIntersect, pt_intersect.cl:92-129 */
static v3 flatTriAndRayIntersect( v3 a, v3 b, v3 c, const ray4* ray, float* t, float tNear ) {
	const float f = fmaxf( 0.0f, tNear - 0.001f );
	const v3 closeOrigin = v3_fma_s( f, ray->dir, ray->origin );
	const v3 edge1 = v3_sub( b, a );
	const v3 edge2 = v3_sub( c, a );
	const v3 tVec = v3_sub( closeOrigin, a );
	const v3 pVec = v3_cross( ray->dir, edge2 );
	const v3 qVec = v3_cross( tVec, edge1 );
	const float invDet = det_rcp( v3_dot( edge1, pVec ) );

	*t = v3_dot( edge2, qVec ) * invDet;

	if( *t >= ray->t || *t < EPSILON5 ) {
		*t = ORC_INF;
		return V3( 0.0f, 0.0f, 0.0f );
	}

	const float u = v3_dot( tVec, pVec ) * invDet;
	const float v = v3_dot( ray->dir, qVec ) * invDet;

	if( u + v > 1.0f || fminf( u, v ) < 0.0f ) {
		*t = ORC_INF;
		return V3( 0.0f, 0.0f, 0.0f );
	}

	*t += f;

	return v3_normalize( v3_cross( edge1, edge2 ) );
}

/* ------------------------------------------------------------------------- */
/* pt_phongtess.cl + its helpers in pt_utils.cl (PHONGTESS == 1)              */
/* ------------------------------------------------------------------------- */

/* solveCubic, pt_utils.cl:108-199: a0 x^3 + a1 x^2 + a2 x + a3 = 0, returns the number of real roots in x[] */
static int solveCubic( float a0, float a1, float a2, float a3, float x[3] ) {
	const float THIRD = 0.3333333333f;
	const float THIRD_HALF = 0.1666666666f;
	float w, p, q, dis, phi;

	if( fabsf( a0 ) > 0.0f ) {
		w = det_div( a1, a0 ) * THIRD;
		p = det_div( a2, a0 ) * THIRD - w * w;
		p = p * p * p;
		q = 0.5f * det_div( a2 * w - a3, a0 ) - w * w * w;
		dis = q * q + p;

		if( dis < 0.0f ) {
			phi = det_acos( det_clamp( det_div( q, det_sqrt( -p ) ), -1.0f, 1.0f ) );
			p = 2.0f * det_pow( -p, THIRD_HALF );

			/* ( phi + 2.0f * M_PI ) * THIRD: M_PI is a double literal, the sum and product are binary64 */
			const float u[3] = {
				p * det_cos( phi * THIRD ) - w,
				p * det_cos( (float) ( ( (double) phi + (double) 2.0f * M_PI_D ) * (double) THIRD ) ) - w,
				p * det_cos( (float) ( ( (double) phi + (double) 4.0f * M_PI_D ) * (double) THIRD ) ) - w
			};

			x[0] = fminf( u[0], fminf( u[1], u[2] ) );
			x[1] = fmaxf( fminf( u[0], u[1] ), fmaxf( fminf( u[0], u[2] ), fminf( u[1], u[2] ) ) );
			x[2] = fmaxf( u[0], fmaxf( u[1], u[2] ) );

			for( int k = 0; k < 3; k++ ) {
				x[k] -= det_div(
					a3 + x[k] * ( a2 + x[k] * ( a1 + x[k] * a0 ) ),
					a2 + x[k] * ( 2.0f * a1 + x[k] * 3.0f * a0 )
				);
			}

			return 3;
		}

		dis = det_sqrt( dis );
		x[0] = det_cbrt( q + dis ) + det_cbrt( q - dis ) - w;
		x[0] -= det_div(
			a3 + x[0] * ( a2 + x[0] * ( a1 + x[0] * a0 ) ),
			a2 + x[0] * ( 2.0f * a1 + x[0] * 3.0f * a0 )
		);

		return 1;
	}
	else if( fabsf( a1 ) > 0.0f ) {
		p = 0.5f * det_div( a2, a1 );
		dis = p * p - det_div( a3, a1 );

		if( dis >= 0.0f ) {
			const float dis_sqrt = det_sqrt( dis );
			x[0] = -p - dis_sqrt;
			x[1] = -p + dis_sqrt;
			x[0] -= det_div( a3 + x[0] * ( a2 + x[0] * a1 ), a2 + x[0] * 2.0f * a1 );
			x[1] -= det_div( a3 + x[1] * ( a2 + x[1] * a1 ), a2 + x[1] * 2.0f * a1 );
			return 2;
		}
	}
	else if( fabsf( a2 ) > 0.0f ) {
		x[0] = det_div( -a3, a2 );
		return 1;
	}

	return 0;
}

/* projectOnPlane, pt_utils.cl:397-399 */
static inline v3 projectOnPlane( v3 q, v3 p, v3 n ) {
	return v3_sub( q, v3_scale( n, v3_dot( v3_sub( q, p ), n ) ) );
}

/* phongTessellation, pt_phongtess.cl:14-26 */
static v3 phongTessellation( v3 P1, v3 P2, v3 P3, v3 N1, v3 N2, v3 N3, float u, float v, float w, float alpha ) {
	const v3 pBary = v3_add( v3_add( v3_scale( P1, u ), v3_scale( P2, v ) ), v3_scale( P3, w ) );
	const v3 pTessellated = v3_add(
		v3_add( v3_scale( projectOnPlane( pBary, P1, N1 ), u ), v3_scale( projectOnPlane( pBary, P2, N2 ), v ) ),
		v3_scale( projectOnPlane( pBary, P3, N3 ), w )
	);

	return v3_add( v3_scale( pBary, 1.0f - alpha ), v3_scale( pTessellated, alpha ) );
}

/* getTriangleNormalS / getTriangleNormal / getTriangleReflectionVec / getPhongTessNormal, pt_utils.cl:231-294 */
static v3 getPhongTessNormal(
	v3 an, v3 bn, v3 cn, v3 rayDir, float u, float v, float w, v3 C1, v3 C2, v3 C3, v3 E12, v3 E20
) {
	const v3 du = v3_add( v3_add( v3_scale( C3, w - u ), v3_scale( v3_sub( C1, C2 ), v ) ), E20 );
	const v3 dv = v3_sub( v3_add( v3_scale( C2, w - v ), v3_scale( v3_sub( C1, C3 ), u ) ), E12 );
	const v3 ns = v3_normalize( v3_cross( du, dv ) );
	const v3 np = v3_normalize( v3_add( v3_add( v3_scale( an, u ), v3_scale( bn, v ) ), v3_scale( cn, w ) ) );
	const v3 r = v3_sub( rayDir, v3_scale( v3_scale( np, 2.0f ), v3_dot( rayDir, np ) ) );

	return ( v3_dot( ns, r ) < 0.0f ) ? ns : np;
}

/* phongTessTriAndRayIntersect, pt_phongtess.cl:56-212 (after Ogaki & Tokuyoshi, "Direct Ray Tracing of Phong
 * Tessellation"); getPlanesFromRay (pt_utils.cl:208-218) and getBestRayDomain (pt_phongtess.cl:35-44) inlined */
static v3 phongTessTriAndRayIntersect(
	v3 P1, v3 P2, v3 P3, v3 N1, v3 N2, v3 N3, const ray4* ray, float* t, float tNear, float tFar, float alpha
) {
	v3 normal = V3( 0.0f, 0.0f, 0.0f );
	*t = ORC_INF;

	const v3 E01 = v3_sub( P2, P1 );
	const v3 E12 = v3_sub( P3, P2 );
	const v3 E20 = v3_sub( P1, P3 );
	const v3 C1 = v3_scale( v3_sub( v3_scale( N2, v3_dot( N2, E01 ) ), v3_scale( N1, v3_dot( N1, E01 ) ) ), alpha );
	const v3 C2 = v3_scale( v3_sub( v3_scale( N3, v3_dot( N3, E12 ) ), v3_scale( N2, v3_dot( N2, E12 ) ) ), alpha );
	const v3 C3 = v3_scale( v3_sub( v3_scale( N1, v3_dot( N1, E20 ) ), v3_scale( N3, v3_dot( N3, E20 ) ) ), alpha );

	const v3 n1 = v3_normalize( v3_cross( ray->origin, ray->dir ) );
	const v3 n2 = v3_normalize( v3_cross( n1, ray->dir ) );
	const float o1 = v3_dot( n1, ray->origin );
	const float o2 = v3_dot( n2, ray->origin );
	const v3 C123 = v3_sub( v3_sub( C1, C2 ), C3 );

	const float a = v3_dot( v3_neg( n1 ), C3 );
	const float b = v3_dot( v3_neg( n1 ), C2 );
	const float c = v3_dot( n1, P3 ) - o1;
	const float d = v3_dot( n1, C123 ) * 0.5f;
	const float e = v3_dot( n1, v3_add( C3, E20 ) ) * 0.5f;
	const float f = v3_dot( n1, v3_sub( C2, E12 ) ) * 0.5f;
	const float l = v3_dot( v3_neg( n2 ), C3 );
	const float m = v3_dot( v3_neg( n2 ), C2 );
	const float n = v3_dot( n2, P3 ) - o2;
	const float o = v3_dot( n2, C123 ) * 0.5f;
	const float p = v3_dot( n2, v3_add( C3, E20 ) ) * 0.5f;
	const float q = v3_dot( n2, v3_sub( C2, E12 ) ) * 0.5f;

	float xs[3] = { -1.0f, -1.0f, -1.0f };
	const float a3 = ( l*m*n + 2.0f*o*p*q ) - ( l*q*q + m*p*p + n*o*o );
	const float a2 = ( a*m*n + l*b*n + l*m*c + 2.0f*( d*p*q + o*e*q + o*p*f ) ) -
	                 ( a*q*q + b*p*p + c*o*o + 2.0f*( l*f*q + m*e*p + n*d*o ) );
	const float a1 = ( a*b*n + a*m*c + l*b*c + 2.0f*( o*e*f + d*e*q + d*p*f ) ) -
	                 ( l*f*f + m*e*e + n*d*d + 2.0f*( a*f*q + b*e*p + c*d*o ) );
	const float a0 = ( a*b*c + 2.0f*d*e*f ) - ( a*f*f + b*e*e + c*d*d );
	const int numCubicRoots = solveCubic( a0, a1, a2, a3, xs );

	if( numCubicRoots == 0 ) {
		return normal;
	}

	float x = 0.0f;
	float determinant = ORC_INF;
	float mA, mB, mC, mD, mE, mF;

	for( int i = 0; i < numCubicRoots; i++ ) {
		mA = a * xs[i] + l;
		mB = b * xs[i] + m;
		mD = d * xs[i] + o;
		const float tmp = mD * mD - mA * mB;
		x = ( determinant > tmp ) ? xs[i] : x;
		determinant = fminf( determinant, tmp );
	}

	if( 0.0f >= determinant ) {
		return normal;
	}

	const v3 ad = V3( fabsf( ray->dir.x ), fabsf( ray->dir.y ), fabsf( ray->dir.z ) );
	int domain = ( ad.y > ad.z ) ? 1 : 2;

	if( ad.x > ad.y ) {
		domain = ( ad.x > ad.z ) ? 0 : 2;
	}

	mA = a * x + l;
	mB = b * x + m;
	mC = c * x + n;
	mD = d * x + o;
	mE = e * x + p;
	mF = f * x + q;

	const int AlessB = fabsf( mA ) < fabsf( mB );
	const float mBorA = AlessB ? mB : mA;
	mA = det_div( mA, mBorA );
	mB = det_div( mB, mBorA );
	mC = det_div( mC, mBorA );
	mD = det_div( mD, mBorA );
	mE = det_div( mE, mBorA );
	mF = det_div( mF, mBorA );

	const float mAorB = AlessB ? mA : mB;
	const float mEorF = AlessB ? 2.0f * mE : 2.0f * mF;
	const float mForE = AlessB ? mF : mE;
	const float ab = AlessB ? a : b;
	const float ba = AlessB ? b : a;
	const float ef = AlessB ? e : f;
	const float fe = AlessB ? f : e;

	const float sqrtAorB = det_sqrt( mD * mD - mAorB );
	const float sqrtC = det_sqrt( mForE * mForE - mC );
	const float lab1 = mD + sqrtAorB;
	const float lab2 = mD - sqrtAorB;
	float lc1 = mForE + sqrtC;
	float lc2 = mForE - sqrtC;

	if( fabsf( mEorF - lab1 * lc1 - lab2 * lc2 ) < fabsf( mEorF - lab1 * lc2 - lab2 * lc1 ) ) {
		const float tmp = lc1;
		lc1 = lc2;
		lc2 = tmp;
	}

	for( int loop = 0; loop < 2; loop++ ) {
		const float g = ( loop == 0 ) ? -lab1 : -lab2;
		const float h = ( loop == 0 ) ? -lc1 : -lc2;
		const float c0 = ab + g * ( 2.0f * d + ba * g );
		const float c1 = 2.0f * ( h * ( d + ba * g ) + ef + fe * g );
		const float c2 = h * ( ba * h + 2.0f * fe ) + c;
		const int numResults = solveCubic( 0.0f, c0, c1, c2, xs );

		for( int i = 0; i < numResults; i++ ) {
			float u = xs[i];
			float v = g * u + h;
			const float w = 1.0f - u - v;

			if( u < 0.0f || v < 0.0f || w < 0.0f ) {
				continue;
			}

			if( !AlessB ) {
				const float tmp = u;
				u = v;
				v = tmp;
			}

			const v3 pTessellated = v3_sub( phongTessellation( P1, P2, P3, N1, N2, N3, u, v, w, alpha ), ray->origin );
			const float num = ( domain == 0 ) ? pTessellated.x : ( domain == 1 ) ? pTessellated.y : pTessellated.z;
			const float den = ( domain == 0 ) ? ray->dir.x : ( domain == 1 ) ? ray->dir.y : ray->dir.z;
			const float tParam = det_div( num, den );

			if( tParam >= fabsf( tNear ) && tParam <= fminf( *t, fminf( ray->t, tFar ) ) ) {
				*t = tParam;
				normal = getPhongTessNormal( N1, N2, N3, ray->dir, u, v, w, C1, C2, C3, E12, E20 );
			}
		}
	}

	return normal;
}

/* checkFaceIntersection, pt_intersect.cl:142-176: with PHONGTESS == 1 a face whose three vertex normals are
 * equal (component-wise ==) still takes the flat test */
static v3 checkFaceIntersection( const ctx_t* c, const ray4* ray, int fIndex, float* t, float tNear, float tFar ) {
	const orc_uint4 fv = c->scene->facesV[fIndex];
	const v3 a = v3_from4( c->scene->vertices[fv.x] );
	const v3 b = v3_from4( c->scene->vertices[fv.y] );
	const v3 cc = v3_from4( c->scene->vertices[fv.z] );

	if( c->cfg->phong_tessellation > 0.0f ) {
		const orc_uint4 fn = c->scene->facesN[fIndex];
		const v3 an = v3_from4( c->scene->normals[fn.x] );
		const v3 bn = v3_from4( c->scene->normals[fn.y] );
		const v3 cn = v3_from4( c->scene->normals[fn.z] );
		const int allEqual = an.x == bn.x && an.y == bn.y && an.z == bn.z && bn.x == cn.x && bn.y == cn.y && bn.z == cn.z;

		if( !allEqual ) {
			return phongTessTriAndRayIntersect( a, b, cc, an, bn, cn, ray, t, tNear, tFar, c->cfg->phong_tessellation );
		}
	}

	return flatTriAndRayIntersect( a, b, cc, ray, t, tNear );
}


/* ------------------------------------------------------------------------- */
/* pt_bvh.cl                                                                 */
/* ------------------------------------------------------------------------- */

/* intersectFace, pt_bvh.cl:10-24 */
static void intersectFace( ctx_t* c, ray4* ray, int faceIndex, float* t, float tNear, float tFar ) {
	const v3 normal = checkFaceIntersection( c, ray, faceIndex, t, tNear, tFar );

	if( ray->t > *t ) {
		ray->normal = normal;
		ray->hitFace = faceIndex;
		ray->t = *t;
	}

	c->dbg_faces += 1.0f;
}

/* intersectFaces, pt_bvh.cl:35-46 */
static void intersectFaces( ctx_t* c, ray4* ray, const orc_bvh_node* node, float tNear, float tFar ) {
	float t = ORC_INF;

	intersectFace( c, ray, (int) node->bbMin.w, &t, tNear, tFar );

	if( node->bbMax.w == -1.0f ) {
		return;
	}

	intersectFace( c, ray, (int) node->bbMax.w, &t, tNear, tFar );
}

/* traverseLights, pt_bvh.cl:54-74 */
static void traverseLights( const ctx_t* c, ray4* ray ) {
	float tNear = 0.0f;
	float tFar = ORC_INF;

	for( int i = 0; i < c->cfg->num_lights; i++ ) {
		const orc_light light = c->scene->lights[i];

		if( light.data.x == 2.0f ) {
			if(
				intersectSphere( ray, v3_from4( light.pos ), light.data.y, &tNear, &tFar ) &&
				tNear < ray->t
			) {
				ray->t = ORC_INF;
				ray->hitFace = -( i + 1 );
			}
		}
	}
}

/* Optional per-node visit histogram (analysis aid for cache studies; NULL = off). */
static uint32_t* g_node_hist = 0;
void orc_debug_set_node_hist( uint32_t* hist ) { g_node_hist = hist; }

/* Analysis aid (scripts/layout_study.py): the sequence of node indices the closest-hit walk visits,
 * appended to log[0 .. capacity); *count is the running length.  Single-threaded use only. */
static int32_t* g_visit_log = 0;
static uint64_t g_visit_cap = 0;
static uint64_t* g_visit_count = 0;
void orc_debug_set_visit_log( int32_t* log, uint64_t capacity, uint64_t* count ) {
	g_visit_log = log;
	g_visit_cap = capacity;
	g_visit_count = count;
}

/* traverse, pt_bvh.cl:82-123 */
static void traverseOrdered( ctx_t* c, ray4* ray );
static void traverseShadowsOrdered( ctx_t* c, ray4* ray );
static inline void noteWalk( uint32_t visits );

static void traverse( ctx_t* c, ray4* ray ) {
	if( c->cfg->traversal != 0 ) {
		traverseOrdered( c, ray );   /* the product's opt-in ray-ordered walk, not the reference's: see below */
		return;
	}

	const v3 invDir = V3( det_rcp( ray->dir.x ), det_rcp( ray->dir.y ), det_rcp( ray->dir.z ) );
	const int numNodes = c->cfg->num_nodes;
	int index = 1;
	uint32_t walkVisits = 0;   /* analysis aid only (orc_debug_set_walk_max) */

	traverseLights( c, ray );

	do {
		c->dbg_nodes += 1.0f;
		walkVisits++;
		if( g_node_hist ) {
			__atomic_fetch_add( &g_node_hist[index], 1u, __ATOMIC_RELAXED );
		}
		if( g_visit_log && *g_visit_count < g_visit_cap ) {
			g_visit_log[( *g_visit_count )++] = index;
		}
		const orc_bvh_node node = c->scene->bvh[index];
		const int currentIndex = index;

		index = ( node.bbMin.w <= -1.0f ) ? (int) node.bbMax.w : currentIndex + 1;

		float tNear = 0.0f;
		float tFar = ORC_INF;

		const int isNodeHit = (
			intersectBox( ray, &invDir, node.bbMin, node.bbMax, &tNear, &tFar ) &&
			tFar > EPSILON5 && ray->t > tNear
		);

		if( !isNodeHit ) {
			continue;
		}

		index = currentIndex + 1;

		if( node.bbMin.w >= 0.0f ) {
			intersectFaces( c, ray, &node, tNear, tFar );
		}
	} while( index > 0 && index < numNodes );

	noteWalk( walkVisits );
}

/* traverseShadows, pt_bvh.cl:133-177 (the bbMin.w == -2 branch is kept although the host never emits -2) */
static void traverseShadows( ctx_t* c, ray4* ray ) {
	if( c->cfg->traversal != 0 ) {
		traverseShadowsOrdered( c, ray );
		return;
	}

	const float tLight = ray->t;
	const v3 invDir = V3( det_rcp( ray->dir.x ), det_rcp( ray->dir.y ), det_rcp( ray->dir.z ) );
	const int numNodes = c->cfg->num_nodes;
	int index = 1;

	traverseLights( c, ray );

	do {
		const orc_bvh_node node = c->scene->bvh[index];
		const int currentIndex = index;

		index = ( node.bbMin.w <= -1.0f ) ? (int) node.bbMax.w : currentIndex + 1;

		float tNear = 0.0f;
		float tFar = ORC_INF;

		const int isNodeHit = (
			intersectBox( ray, &invDir, node.bbMin, node.bbMax, &tNear, &tFar ) &&
			tFar > EPSILON5
		);

		if( !isNodeHit ) {
			continue;
		}

		index = currentIndex + 1;

		if( node.bbMin.w == -2.0f ) {
			index++;
		}

		if( node.bbMin.w >= 0.0f ) {
			intersectFaces( c, ray, &node, tNear, tFar );

			if( ray->t < tLight ) {
				break;
			}
		}
	} while( index > 0 && index < numNodes );
}


/* ------------------------------------------------------------------------- */
/* Ray-ordered walk — NOT in the reference                                   */
/* ------------------------------------------------------------------------- */

/* The reference walks its flat tree in ONE order: a hit continues at index + 1 (pt_bvh.cl:102,112) and the builder
 * puts the child with the bigger surface area there, whatever the ray (accelstructures/BVH.cpp:335-343).  The product
 * has an opt-in mode (pbr_config.traversal) that walks the SAME flat tree — same boxes, same leaves, same arithmetic per
 * visit (intersectBox, the hit condition of pt_bvh.cl:107-110, intersectFaces) — with the children of every container
 * ordered along the ray.  This is its CPU statement, so that the HIP path has something to be bit-identical to; what it
 * owes the reference is checked separately (tests: the image against the reference-order image within SURVEY 8(c)'s
 * tolerance; only exact ties of the closest hit can differ, because intersectFace keeps the first of two equal t).
 *
 * The tree behind the flat array: a leaf ends at index + 1; a container i ends at its miss link when that is > i, else
 * where its parent ends (the root: N); its children are c0 = i + 1, c1 = end( c0 ), ... while < end( i ) (the flattening
 * drops nodes, PathTracer.cpp:250-256, so a container can have more than two).
 *
 * Scheme 1 (six orders): order k = 2 * axis + negative, axis = the ray direction's dominant axis (x before y before z
 * on ties), negative = dir[axis] < 0.  A child's key on an axis is bbMin[axis] + bbMax[axis] in binary32.  Order
 * ( axis, + ): insertion sort of the DFS child list, a child moves in front of its predecessor while its key is SMALLER;
 * ( axis, - ): while its key is GREATER.  (Stated as an algorithm so that NaN keys and ties have one outcome.)
 * Scheme 2 (eight orders): k = sign bits of dir ( x | y << 1 | z << 2 ); every container sorts on ITS axis — the one on
 * which its children's keys spread furthest (max - min; x before y before z on ties) — ascending if that sign bit is 0.
 *
 * Links of order k: hit( container ) = its first child in that order; next( child j ) = child j + 1, the last child's
 * next = next( parent ); next( root ) = -1 (end).  A missed container and every leaf continue at next.
 * links[( k * N + i ) * 2 + {0 hit, 1 miss}], first[k] = hit_k( root ). */
int orc_walk_order_count( int scheme ) {
	return ( scheme == 1 ) ? 6 : ( scheme == 2 ) ? 8 : 0;
}

static inline float nodeKey( const orc_bvh_node* n, int axis ) {
	const float lo = ( axis == 0 ) ? n->bbMin.x : ( axis == 1 ) ? n->bbMin.y : n->bbMin.z;
	const float hi = ( axis == 0 ) ? n->bbMax.x : ( axis == 1 ) ? n->bbMax.y : n->bbMax.z;
	return lo + hi;
}

int orc_build_walk_orders( const orc_bvh_node* bvh, int numNodes, int scheme, int32_t* links, int32_t* first ) {
	const int K = orc_walk_order_count( scheme );
	const size_t N = (size_t) numNodes;

	if( K == 0 || numNodes < 2 ) {
		return -1;
	}

	int32_t* end = (int32_t*) malloc( sizeof( int32_t ) * N );
	int32_t* stack = (int32_t*) malloc( sizeof( int32_t ) * N );
	int32_t* kids = (int32_t*) malloc( sizeof( int32_t ) * N );
	int32_t* sorted = (int32_t*) malloc( sizeof( int32_t ) * N );
	int32_t* rootNext = (int32_t*) malloc( sizeof( int32_t ) * (size_t) K );
	int top = 0;

	/* end( i ): one pass with the open containers on a stack */
	for( int i = 0; i < numNodes; i++ ) {
		while( top > 0 && i >= end[stack[top - 1]] ) {
			top--;
		}

		if( bvh[i].bbMin.w >= 0.0f ) {
			end[i] = i + 1;
		}
		else {
			const int link = (int) bvh[i].bbMax.w;
			end[i] = ( link > i ) ? link : ( ( top > 0 ) ? end[stack[top - 1]] : numNodes );
			stack[top++] = i;
		}
	}

	for( int k = 0; k < K; k++ ) {
		rootNext[k] = -1;
	}

	for( int i = 0; i < numNodes; i++ ) {
		if( bvh[i].bbMin.w >= 0.0f ) {
			for( int k = 0; k < K; k++ ) {
				int32_t* L = links + ( (size_t) k * N + (size_t) i ) * 2;
				L[0] = L[1];   /* a leaf continues at next whether it is hit or not; next was written by its parent */
			}
			continue;
		}

		int n = 0;

		for( int c = i + 1; c < end[i]; c = end[c] ) {
			kids[n++] = c;
		}

		/* scheme 2: the axis on which the children's keys spread furthest */
		int ownAxis = 0;

		if( scheme == 2 ) {
			float best = -1.0f;

			for( int a = 0; a < 3; a++ ) {
				float lo = ORC_INF, hi = -ORC_INF;

				for( int j = 0; j < n; j++ ) {
					const float key = nodeKey( &bvh[kids[j]], a );
					lo = ( key < lo ) ? key : lo;
					hi = ( key > hi ) ? key : hi;
				}

				const float spread = hi - lo;

				if( spread > best ) {
					best = spread;
					ownAxis = a;
				}
			}
		}

		for( int k = 0; k < K; k++ ) {
			const int axis = ( scheme == 1 ) ? ( k >> 1 ) : ownAxis;
			const int descending = ( scheme == 1 ) ? ( k & 1 ) : ( ( k >> ownAxis ) & 1 );

			for( int j = 0; j < n; j++ ) {
				const float key = nodeKey( &bvh[kids[j]], axis );
				int at = j;

				while( at > 0 ) {
					const float prev = nodeKey( &bvh[sorted[at - 1]], axis );

					if( !( descending ? ( key > prev ) : ( key < prev ) ) ) {
						break;
					}

					sorted[at] = sorted[at - 1];
					at--;
				}

				sorted[at] = kids[j];
			}

			int32_t* L = links + ( (size_t) k * N + (size_t) i ) * 2;
			const int32_t next = ( i == 0 ) ? rootNext[k] : L[1];

			L[0] = ( n > 0 ) ? sorted[0] : next;
			L[1] = next;

			for( int j = 0; j < n; j++ ) {
				links[( (size_t) k * N + (size_t) sorted[j] ) * 2 + 1] = ( j + 1 < n ) ? sorted[j + 1] : next;
			}

			if( i == 0 ) {
				first[k] = L[0];
			}
		}
	}

	free( end ); free( stack ); free( kids ); free( sorted ); free( rootNext );
	return 0;
}

static inline int walkOrderOf( int scheme, v3 d ) {   /* (scheme 3, the analysis aid above, walks its shadow rays in six orders) */
	if( scheme == 2 ) {
		return ( d.x < 0.0f ) | ( ( d.y < 0.0f ) << 1 ) | ( ( d.z < 0.0f ) << 2 );
	}

	const float ax = fabsf( d.x ), ay = fabsf( d.y ), az = fabsf( d.z );
	const int axis = ( ax >= ay && ax >= az ) ? 0 : ( ( ay >= az ) ? 1 : 2 );
	const float along = ( axis == 0 ) ? d.x : ( axis == 1 ) ? d.y : d.z;
	return 2 * axis + ( along < 0.0f );
}

/* Analysis aid: the longest single walk (node visits) seen since it was set; NULL = off. */
static uint32_t* g_walk_max = 0;
void orc_debug_set_walk_max( uint32_t* slot ) { g_walk_max = slot; }

/* Analysis aid (scripts/wave_orders.py, round 6), single-threaded runs only: every closest-hit walk of the ray-ordered mode
 * appends one word — its order k | depth of the path at that walk << 4 | its node visits << 12 — so that the orders the
 * paths of a wave would walk can be counted on the CPU. */
static uint32_t* g_walk_log = 0;
static uint32_t g_walk_log_cap = 0;
static uint32_t* g_walk_log_count = 0;
static uint32_t g_walk_depth = 0;
void orc_debug_set_walk_log( uint32_t* log, uint32_t cap, uint32_t* count ) { g_walk_log = log; g_walk_log_cap = cap; g_walk_log_count = count; }

static inline void noteWalk( uint32_t visits ) {
	if( g_walk_max ) {
		uint32_t seen = __atomic_load_n( g_walk_max, __ATOMIC_RELAXED );

		while( visits > seen && !__atomic_compare_exchange_n( g_walk_max, &seen, visits, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED ) ) {
		}
	}
}

/* Analysis aid, NOT a mode of the product and not the reference's algorithm (cfg.traversal == 3; pbr_configure rejects it):
 * the walk a STACK would allow — at every hit container all children are box-tested and the hit ones visited nearest
 * first (by their boxes' entry distance tNear along THIS ray), a child whose tNear is no longer below ray.t when its turn
 * comes is dropped without another test.  Its node count (box tests) is the bound the stackless ordered walks are measured
 * against in profiles/r05/experiments/traversal_order.txt (VERDICT r04 quoted 0.57x / 0.72x / 0.93x for it).  Uses the
 * scheme-1 tables only to enumerate a container's children ( hit = first child, next = sibling chain ). */
static void traverseNearFirst( ctx_t* c, ray4* ray ) {
	const v3 invDir = V3( det_rcp( ray->dir.x ), det_rcp( ray->dir.y ), det_rcp( ray->dir.z ) );
	const int32_t* L = c->scene->walk_links;   /* order 0 of scheme 1: any order enumerates the children */
	enum { MAXSTACK = 4096 };
	int32_t stackNode[MAXSTACK];
	float stackNear[MAXSTACK], stackFar[MAXSTACK];
	int top = 0;
	uint32_t visits = 0;
	int container = 0;   /* the root: never tested itself (pt_bvh.cl:84) */

	traverseLights( c, ray );

	for( ;; ) {
		/* test the children of `container`, push the hit ones farthest first */
		int32_t kids[64];
		float kNear[64], kFar[64];
		int n = 0;
		const int32_t stop = L[2 * container + 1];

		for( int32_t child = L[2 * container]; child > 0 && child != stop && n < 64; child = L[2 * child + 1] ) {
			const orc_bvh_node node = c->scene->bvh[child];
			float tNear = 0.0f, tFar = ORC_INF;
			c->dbg_nodes += 1.0f;
			visits++;

			if( intersectBox( ray, &invDir, node.bbMin, node.bbMax, &tNear, &tFar ) && tFar > EPSILON5 && ray->t > tNear ) {
				kids[n] = child; kNear[n] = tNear; kFar[n] = tFar; n++;
			}
		}

		for( int i = 1; i < n; i++ ) {   /* insertion sort, farthest first */
			const int32_t k = kids[i]; const float a = kNear[i], b = kFar[i];
			int j = i;
			while( j > 0 && kNear[j - 1] < a ) { kids[j] = kids[j - 1]; kNear[j] = kNear[j - 1]; kFar[j] = kFar[j - 1]; j--; }
			kids[j] = k; kNear[j] = a; kFar[j] = b;
		}

		for( int i = 0; i < n && top < MAXSTACK; i++ ) {
			stackNode[top] = kids[i]; stackNear[top] = kNear[i]; stackFar[top] = kFar[i]; top++;
		}

		/* next: the nearest pending node that can still hold a closer hit */
		container = -1;

		while( top > 0 ) {
			top--;

			if( !( ray->t > stackNear[top] ) ) {
				continue;
			}

			const orc_bvh_node node = c->scene->bvh[stackNode[top]];

			if( node.bbMin.w >= 0.0f ) {
				intersectFaces( c, ray, &node, stackNear[top], stackFar[top] );
			}
			else {
				container = stackNode[top];
				break;
			}
		}

		if( container < 0 ) {
			break;
		}
	}

	noteWalk( visits );
}

/* traverse (pt_bvh.cl:82-123) with the successors of the ray's order: per visit the reference's statements */
static void traverseOrdered( ctx_t* c, ray4* ray ) {
	if( c->cfg->traversal == 3 ) {
		traverseNearFirst( c, ray );
		return;
	}

	const v3 invDir = V3( det_rcp( ray->dir.x ), det_rcp( ray->dir.y ), det_rcp( ray->dir.z ) );
	const size_t N = (size_t) c->cfg->num_nodes;
	const int k = walkOrderOf( c->cfg->traversal, ray->dir );
	const int32_t* L = c->scene->walk_links + (size_t) k * N * 2;
	int index = c->scene->walk_first[k];
	uint32_t visits = 0;

	traverseLights( c, ray );

	while( index > 0 ) {
		c->dbg_nodes += 1.0f;
		visits++;
		if( g_node_hist ) {
			__atomic_fetch_add( &g_node_hist[(size_t) k * N + (size_t) index], 1u, __ATOMIC_RELAXED );
		}
		if( g_visit_log && *g_visit_count < g_visit_cap ) {
			g_visit_log[( *g_visit_count )++] = index;
		}
		const orc_bvh_node node = c->scene->bvh[index];
		const int currentIndex = index;

		index = L[2 * currentIndex + 1];

		float tNear = 0.0f;
		float tFar = ORC_INF;

		const int isNodeHit = (
			intersectBox( ray, &invDir, node.bbMin, node.bbMax, &tNear, &tFar ) &&
			tFar > EPSILON5 && ray->t > tNear
		);

		if( !isNodeHit ) {
			continue;
		}

		index = L[2 * currentIndex];

		if( node.bbMin.w >= 0.0f ) {
			intersectFaces( c, ray, &node, tNear, tFar );
		}
	}

	if( g_walk_log && *g_walk_log_count < g_walk_log_cap ) {
		g_walk_log[( *g_walk_log_count )++] = (uint32_t) k | ( ( g_walk_depth & 255u ) << 4 ) | ( visits << 12 );
	}

	noteWalk( visits );
}

/* traverseShadows (pt_bvh.cl:133-177) likewise */
static void traverseShadowsOrdered( ctx_t* c, ray4* ray ) {
	const float tLight = ray->t;
	const v3 invDir = V3( det_rcp( ray->dir.x ), det_rcp( ray->dir.y ), det_rcp( ray->dir.z ) );
	const size_t N = (size_t) c->cfg->num_nodes;
	const int k = walkOrderOf( c->cfg->traversal, ray->dir );
	const int32_t* L = c->scene->walk_links + (size_t) k * N * 2;
	int index = c->scene->walk_first[k];

	traverseLights( c, ray );

	while( index > 0 ) {
		const orc_bvh_node node = c->scene->bvh[index];
		const int currentIndex = index;

		index = L[2 * currentIndex + 1];

		float tNear = 0.0f;
		float tFar = ORC_INF;

		const int isNodeHit = (
			intersectBox( ray, &invDir, node.bbMin, node.bbMax, &tNear, &tFar ) &&
			tFar > EPSILON5
		);

		if( !isNodeHit ) {
			continue;
		}

		index = L[2 * currentIndex];

		if( node.bbMin.w >= 0.0f ) {
			intersectFaces( c, ray, &node, tNear, tFar );

			if( ray->t < tLight ) {
				break;
			}
		}
	}
}


/* ------------------------------------------------------------------------- */
/* pathtracing.cl                                                            */
/* ------------------------------------------------------------------------- */

/* initRay, pathtracing.cl:25-48 */
static ray4 initRay(
	const ctx_t* c, int px, int py, float pxDim, const orc_camera* cam,
	float* seed, float tFocus, float tObject
) {
	const v3 cu = v3_from4( cam->u );
	const v3 cv = v3_from4( cam->v );
	const float W = (float) c->cfg->width;
	const float H = (float) c->cfg->height;
	const float fx = 2.0f * (float) px;
	const float fy = 2.0f * (float) py;

	/* cam.u - IMG_WIDTH * cam.u + 2.0f * pos.x * cam.u + cam.v - IMG_HEIGHT * cam.v + 2.0f * pos.y * cam.v */
	v3 inner = v3_sub( cu, v3_scale( cu, W ) );
	inner = v3_add( inner, v3_scale( cu, fx ) );
	inner = v3_add( inner, cv );
	inner = v3_sub( inner, v3_scale( cv, H ) );
	inner = v3_add( inner, v3_scale( cv, fy ) );

	const float s = pxDim * 0.5f;
	const v3 initialRay = v3_add( v3_from4( cam->w ), v3_scale( inner, s ) );

	ray4 ray;
	ray.t = ORC_INF;
	ray.origin = v3_from4( cam->eye );
	ray.dir = v3_normalize( initialRay );
	ray.normal = V3( 0.0f, 0.0f, 0.0f );
	ray.hitFace = 0;

	antiAliasing( c, &ray, pxDim, seed );

	if( tFocus >= 0.0f && tObject >= 0.0f ) {
		depthOfField( &ray, cam, tObject, tFocus, seed );
	}

	return ray;
}

/* updateColor, pathtracing.cl:89-178 */
static void updateColor(
	const ctx_t* c, const ray4* ray, const ray4* newRay, const mtl_t* mtl,
	const ray4* lightRay, v3 lightRaySource, uint32_t* secondaryPaths,
	v3* color, v3* finalColor
) {
	const float d = mtl->d;

	if( c->cfg->brdf == 0 ) {
		float brdf, pdf, u;

		if( c->cfg->shadow_rays == 1 && lightRaySource.x >= 0.0f ) {
			brdf = brdfSchlick( mtl, ray, lightRay, &ray->normal, &u, &pdf );

			if( fabsf( pdf ) > 0.00001f ) {
				brdf *= fmaxf( v3_dot( ray->normal, lightRay->dir ), 0.0f );
				brdf = det_div( brdf, pdf );

				/* *finalColor += *color * lightRaySource * mtl->rgbDiff *
				 *   ( fresnel4( u, mtl->rgbSpec ) * brdf * mtl->data.s0 + ( 1.0f - mtl->data.s0 ) ) */
				const v3 f4 = V3( fresnel( u, mtl->rgbSpec.x ), fresnel( u, mtl->rgbSpec.y ), fresnel( u, mtl->rgbSpec.z ) );
				const v3 k = V3( f4.x * brdf * d + ( 1.0f - d ), f4.y * brdf * d + ( 1.0f - d ), f4.z * brdf * d + ( 1.0f - d ) );
				const v3 add = v3_mul( v3_mul( v3_mul( *color, lightRaySource ), mtl->rgbDiff ), k );
				*finalColor = v3_add( *finalColor, add );

				*secondaryPaths += 1;
			}
		}

		brdf = brdfSchlick( mtl, ray, newRay, &ray->normal, &u, &pdf );
		brdf *= fmaxf( v3_dot( ray->normal, newRay->dir ), 0.0f );
		brdf = det_div( brdf, pdf );

		/* *color *= mtl->rgbDiff * ( fresnel4( u, mtl->rgbSpec ) * brdf * d + ( 1 - d ) ) */
		const v3 f4 = V3( fresnel( u, mtl->rgbSpec.x ), fresnel( u, mtl->rgbSpec.y ), fresnel( u, mtl->rgbSpec.z ) );
		const v3 k = V3( f4.x * brdf * d + ( 1.0f - d ), f4.y * brdf * d + ( 1.0f - d ), f4.z * brdf * d + ( 1.0f - d ) );
		*color = v3_mul( *color, v3_mul( mtl->rgbDiff, k ) );
	}
	else {
		float brdfDiff, brdfSpec, pdf, dotHK1;

		if( c->cfg->shadow_rays == 1 && lightRaySource.x >= 0.0f ) {
			brdfShirleyAshikhmin(
				mtl->nu, mtl->nv, mtl->Rs, mtl->Rd,
				ray, lightRay, &ray->normal, &brdfSpec, &brdfDiff, &dotHK1, &pdf
			);

			if( fabsf( pdf ) > 0.00001f ) {
				brdfSpec = det_div( brdfSpec, pdf );
				brdfDiff = det_div( brdfDiff, pdf );

				const float fr = fresnel( dotHK1, mtl->Rs );
				const v3 brdf_s = v3_scale( v3_scale( mtl->rgbSpec, brdfSpec ), fr );
				const v3 brdf_d = v3_scale( v3_scale( mtl->rgbDiff, brdfDiff ), 1.0f - mtl->Rs );

				v3 bc = v3_add( brdf_s, brdf_d );
				bc = V3( bc.x * d + ( 1.0f - d ), bc.y * d + ( 1.0f - d ), bc.z * d + ( 1.0f - d ) );
				const float maxRGB = det_max( 1.0f, det_max( bc.x, det_max( bc.y, bc.z ) ) );
				bc = V3( bc.x / maxRGB, bc.y / maxRGB, bc.z / maxRGB );

				/* *finalColor += clamp( brdfColor, 0, 1 ) * lightRaySource * d + ( 1 - d ) */
				const v3 cl = V3( det_clamp( bc.x, 0.0f, 1.0f ), det_clamp( bc.y, 0.0f, 1.0f ), det_clamp( bc.z, 0.0f, 1.0f ) );
				const v3 add = V3(
					cl.x * lightRaySource.x * d + ( 1.0f - d ),
					cl.y * lightRaySource.y * d + ( 1.0f - d ),
					cl.z * lightRaySource.z * d + ( 1.0f - d )
				);
				*finalColor = v3_add( *finalColor, add );

				*secondaryPaths += 1;
			}
		}

		brdfShirleyAshikhmin(
			mtl->nu, mtl->nv, mtl->Rs, mtl->Rd,
			ray, newRay, &ray->normal, &brdfSpec, &brdfDiff, &dotHK1, &pdf
		);

		brdfSpec = det_div( brdfSpec, pdf );
		brdfDiff = det_div( brdfDiff, pdf );

		const float fr = fresnel( dotHK1, mtl->Rs );
		const v3 brdf_s = v3_scale( v3_scale( mtl->rgbSpec, brdfSpec ), fr );
		const v3 brdf_d = v3_scale( v3_scale( mtl->rgbDiff, brdfDiff ), 1.0f - mtl->Rs );

		v3 bc = v3_add( brdf_s, brdf_d );
		bc = V3( bc.x * d + ( 1.0f - d ), bc.y * d + ( 1.0f - d ), bc.z * d + ( 1.0f - d ) );
		const float maxRGB = det_max( 1.0f, det_max( bc.x, det_max( bc.y, bc.z ) ) );
		bc = V3( bc.x / maxRGB, bc.y / maxRGB, bc.z / maxRGB );

		*color = v3_mul( *color, V3( det_clamp( bc.x, 0.0f, 1.0f ), det_clamp( bc.y, 0.0f, 1.0f ), det_clamp( bc.z, 0.0f, 1.0f ) ) );
	}
}

/* shadowRayTest, pathtracing.cl:188-199 */
static void shadowRayTest( ctx_t* c, const ray4* ray, ray4* lightRay, v3* lightRaySource ) {
	const v3 lpos = v3_from4( c->scene->lights[0].pos );
	lightRay->origin = v3_fma_s( ray->t, ray->dir, ray->origin );
	lightRay->dir = v3_normalize( v3_sub( lpos, lightRay->origin ) );
	const v3 dl = v3_sub( lpos, lightRay->origin );
	const float tLight = det_sqrt( v3_dot( dl, dl ) );
	lightRay->t = tLight;

	traverseShadows( c, lightRay );

	if( lightRay->t >= tLight ) {
		*lightRaySource = v3_from4( c->scene->lights[0].rgb );
	}
}

/* kernel pathTracing for one pixel, pathtracing.cl:207-334 */
static void pathTracingPixel(
	ctx_t* c, int px, int py, float seed, float pixelWeight, float pxDim,
	const orc_camera* cam, const float* imageIn, float* imageOut, float* imageDebug,
	uint64_t* nPaths
) {
	const orc_config* cfg = c->cfg;
	const int W = cfg->width;
	const size_t pixOff = ( (size_t) py * (size_t) W + (size_t) px ) * 4;
	v3 finalColor = V3( 0.0f, 0.0f, 0.0f );

	c->dbg_faces = 0.0f;
	c->dbg_nodes = 0.0f;

	float focus = 0.0f;
	float prevFocusObj = -1.0f, prevFocusFocus = -1.0f;

	/* getPreviousFocus, pathtracing.cl:58-65 (CLK_ADDRESS_CLAMP_TO_EDGE) */
	if( cam->focusPoint[0] >= 0 && cam->focusPoint[1] >= 0 ) {
		int fx = cam->focusPoint[0], fy = cam->focusPoint[1];
		fx = ( fx > W - 1 ) ? W - 1 : fx;
		fy = ( fy > cfg->height - 1 ) ? cfg->height - 1 : fy;
		prevFocusObj = imageIn[pixOff + 3];
		prevFocusFocus = imageIn[( (size_t) fy * (size_t) W + (size_t) fx ) * 4 + 3];
	}

	int addDepth;
	uint32_t secondaryPaths = 1;

	for( uint32_t sample = 0; sample < (uint32_t) cfg->samples; sample++ ) {
		v3 color = V3( 1.0f, 1.0f, 1.0f );
		v3 light = V3( -1.0f, -1.0f, -1.0f );

		ray4 ray = initRay( c, px, py, pxDim, cam, &seed, prevFocusFocus, prevFocusObj );
		int depthAdded = 0;
		*nPaths += 1;

		for( uint32_t depth = 0; depth < (uint32_t) ( cfg->max_depth + depthAdded ); depth++ ) {
			g_walk_depth = depth;
			traverse( c, &ray );

			focus = ( sample + depth == 0 ) ? ray.t : focus;

			if( ray.t == ORC_INF ) {
				if( ray.hitFace < 0 ) {
					light = v3_from4( c->scene->lights[-( ray.hitFace + 1 )].rgb );
				}
				else {
					light = V3( cfg->sky_light[0], cfg->sky_light[1], cfg->sky_light[2] );
				}
				break;
			}

			const mtl_t mtl = load_material( c, c->scene->facesV[ray.hitFace].w );
			c->n_hits += 1;

			addDepth = extendDepth( c, &mtl, &seed );

			if( mtl.d == 1.0f && !addDepth && depth == (uint32_t) ( cfg->max_depth + depthAdded - 1 ) ) {
				break;
			}

			seed += ray.t;

			v3 lightRaySource = V3( -1.0f, -1.0f, -1.0f );
			ray4 lightRay;
			memset( &lightRay, 0, sizeof( lightRay ) );
			lightRay.t = ORC_INF;

			if( cfg->shadow_rays == 1 && cfg->num_lights > 0 && mtl.d > 0.0f ) {
				shadowRayTest( c, &ray, &lightRay, &lightRaySource );
			}

			ray4 newRay = getNewRay( c, &ray, &mtl, &seed, &addDepth );

			if( v3_dot( ray.normal, v3_neg( ray.dir ) ) <= 0.0f ) {
				ray.normal = v3_neg( ray.normal );
			}

			updateColor( c, &ray, &newRay, &mtl, &lightRay, lightRaySource, &secondaryPaths, &color, &finalColor );

			depthAdded += ( addDepth && depthAdded < cfg->max_added_depth );

			const float maxValColor = fmaxf( color.x, fmaxf( color.y, color.z ) );

			if( russianRoulette( (int) depth, depthAdded, maxValColor, &seed ) ) {
				break;
			}

			ray = newRay;
		}

		if( light.x > -1.0f ) {
			color = v3_mul( color, light );
			finalColor = v3_add( finalColor, color );
		}
	}

	const float sp = (float) secondaryPaths;
	finalColor = V3( finalColor.x / sp, finalColor.y / sp, finalColor.z / sp );

	if( cfg->samples > 1 ) {
		const float ns = (float) cfg->samples;
		finalColor = V3( finalColor.x / ns, finalColor.y / ns, finalColor.z / ns );
	}

	/* setColors, pt_rgb.cl:9-21: mix( finalColor, imagePixel, pixelWeight ); .w = focus */
	imageOut[pixOff + 0] = finalColor.x + ( imageIn[pixOff + 0] - finalColor.x ) * pixelWeight;
	imageOut[pixOff + 1] = finalColor.y + ( imageIn[pixOff + 1] - finalColor.y ) * pixelWeight;
	imageOut[pixOff + 2] = finalColor.z + ( imageIn[pixOff + 2] - finalColor.z ) * pixelWeight;
	imageOut[pixOff + 3] = focus;

	/* writeDebugImage, pathtracing.cl:73-78 */
	if( imageDebug ) {
		imageDebug[pixOff + 0] = c->dbg_faces / 1082.0f;
		imageDebug[pixOff + 1] = c->dbg_nodes / 1265.0f;
		imageDebug[pixOff + 2] = 0.0f;
		imageDebug[pixOff + 3] = 0.0f;
	}
}


/* ------------------------------------------------------------------------- */
/* Entry points                                                              */
/* ------------------------------------------------------------------------- */

void orc_render_frame(
	const orc_scene* scene, const orc_config* cfg, const orc_camera* cam,
	float seed, float pixelWeight, float pxDim,
	const float* imageIn, float* imageOut, float* imageDebug,
	int y0, int y1, int threads, orc_counters* counters
) {
	const int W = cfg->width;
	const int bands = ( y1 - y0 + 7 ) / 8;
	uint64_t tot_nodes = 0, tot_tris = 0, tot_hits = 0, tot_paths = 0;

	if( threads < 1 ) {
		threads = 1;
	}

	#pragma omp parallel for schedule( dynamic, 1 ) num_threads( threads ) \
		reduction( + : tot_nodes, tot_tris, tot_hits, tot_paths )
	for( int band = 0; band < bands; band++ ) {
		ctx_t c;
		c.scene = scene;
		c.cfg = cfg;
		c.n_hits = 0;
		uint64_t nodes = 0, tris = 0, paths = 0;
		const int ya = y0 + band * 8;
		const int yb = ( ya + 8 < y1 ) ? ya + 8 : y1;

		for( int y = ya; y < yb; y++ ) {
			for( int x = 0; x < W; x++ ) {
				pathTracingPixel( &c, x, y, seed, pixelWeight, pxDim, cam, imageIn, imageOut, imageDebug, &paths );
				nodes += (uint64_t) c.dbg_nodes;
				tris += (uint64_t) c.dbg_faces;
			}
		}

		tot_nodes += nodes;
		tot_tris += tris;
		tot_hits += c.n_hits;
		tot_paths += paths;
	}

	if( counters ) {
		counters->nodes += tot_nodes;
		counters->tris += tot_tris;
		counters->hits += tot_hits;
		counters->paths += tot_paths;
	}
}

void orc_trace_rays(
	const orc_scene* scene, const orc_config* cfg, const float* rays, int n,
	float* out_t, int32_t* out_face, float* out_normal, uint32_t* out_counts
) {
	ctx_t c;
	c.scene = scene;
	c.cfg = cfg;
	c.n_hits = 0;

	for( int i = 0; i < n; i++ ) {
		ray4 ray;
		ray.origin = V3( rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2] );
		ray.dir = V3( rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5] );
		ray.normal = V3( 0.0f, 0.0f, 0.0f );
		ray.t = ORC_INF;
		ray.hitFace = 0;
		c.dbg_faces = 0.0f;
		c.dbg_nodes = 0.0f;

		traverse( &c, &ray );

		out_t[i] = ray.t;
		out_face[i] = ray.hitFace;
		out_normal[i * 3 + 0] = ray.normal.x;
		out_normal[i * 3 + 1] = ray.normal.y;
		out_normal[i * 3 + 2] = ray.normal.z;
		out_counts[i * 2 + 0] = (uint32_t) c.dbg_nodes;
		out_counts[i * 2 + 1] = (uint32_t) c.dbg_faces;
	}
}

void orc_math( int op, const float* x, const float* y, int n, float* out ) {
	for( int i = 0; i < n; i++ ) {
		switch( op ) {
			case 0: out[i] = det_sin( x[i] ); break;
			case 1: out[i] = det_cos( x[i] ); break;
			case 2: out[i] = det_tan( x[i] ); break;
			case 3: out[i] = det_acos( x[i] ); break;
			case 4: out[i] = det_atan( x[i] ); break;
			case 5: out[i] = det_pow( x[i], y[i] ); break;
			case 6: out[i] = det_fract( det_sin( x[i] ) * 43758.5453123f ); break;
			default: out[i] = 0.0f; break;
		}
	}
}

static mtl_t mtl_from_wire( int brdf, const void* mtl ) {
	orc_scene s;
	orc_config cfg;
	ctx_t c;
	memset( &s, 0, sizeof( s ) );
	memset( &cfg, 0, sizeof( cfg ) );
	s.materials = mtl;
	cfg.brdf = brdf;
	c.scene = &s;
	c.cfg = &cfg;
	return load_material( &c, 0 );
}

void orc_brdf_eval( int brdf, const void* mtl, const float* in, int n, float* out ) {
	const mtl_t m = mtl_from_wire( brdf, mtl );

	for( int i = 0; i < n; i++ ) {
		const float* p = in + (size_t) i * 16;
		ray4 rOut, rIn;
		memset( &rOut, 0, sizeof( rOut ) );
		memset( &rIn, 0, sizeof( rIn ) );
		rOut.dir = V3( p[0], p[1], p[2] );
		rIn.dir = V3( p[3], p[4], p[5] );
		const v3 normal = V3( p[6], p[7], p[8] );
		float* o = out + (size_t) i * 4;

		if( brdf == 0 ) {
			float u, pdf;
			const float b = brdfSchlick( &m, &rOut, &rIn, &normal, &u, &pdf );
			o[0] = b; o[1] = u; o[2] = pdf; o[3] = 0.0f;
		}
		else {
			float spec, diff, dotHK1, pdf;
			brdfShirleyAshikhmin( m.nu, m.nv, m.Rs, m.Rd, &rOut, &rIn, &normal, &spec, &diff, &dotHK1, &pdf );
			o[0] = spec; o[1] = diff; o[2] = dotHK1; o[3] = pdf;
		}
	}
}

void orc_new_ray( int brdf, const void* mtl, const float* in, int n, float* out ) {
	const mtl_t m = mtl_from_wire( brdf, mtl );
	orc_config cfg;
	ctx_t c;
	memset( &cfg, 0, sizeof( cfg ) );
	cfg.brdf = brdf;
	c.scene = 0;
	c.cfg = &cfg;

	for( int i = 0; i < n; i++ ) {
		const float* p = in + (size_t) i * 12;
		ray4 ray;
		ray.origin = V3( p[0], p[1], p[2] );
		ray.dir = V3( p[3], p[4], p[5] );
		ray.normal = V3( p[6], p[7], p[8] );
		ray.t = p[9];
		ray.hitFace = 0;
		float seed = p[10];
		int addDepth = 0;

		const ray4 nr = getNewRay( &c, &ray, &m, &seed, &addDepth );
		float* o = out + (size_t) i * 8;
		o[0] = nr.origin.x; o[1] = nr.origin.y; o[2] = nr.origin.z;
		o[3] = nr.dir.x; o[4] = nr.dir.y; o[5] = nr.dir.z;
		o[6] = seed;
		o[7] = (float) addDepth;
	}
}
