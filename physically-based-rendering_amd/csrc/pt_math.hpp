// Device math for the path tracer: gfx950 implementations of the deterministic definitions
// DESIGN.md §"Deterministic math" gives for the OpenCL builtins the reference kernels call
// (native_sin/cos/tan/recip/divide/sqrt, fast_normalize, dot, cross, fract, mix, clamp, max,
// pow, acos, atan).  IEEE binary32 / binary64 operations only, contraction OFF for the whole
// translation unit (-ffp-contract=off); every fused multiply-add below is explicit.
//
// PT_ARITH_NATIVE (pt_flavour.hpp; pbr_config.arith = PBR_ARITH_NATIVE): what the reference literally asks its device
// for — native_sin / native_cos / native_tan / native_recip / native_divide / native_sqrt / fast_normalize
// (pt_utils.cl:39-44, pt_brdf.cl:306-321, pt_intersect.cl:104, pt_bvh.cl:83) — as the gfx950 instructions: v_sin_f32 /
// v_cos_f32 on fract( x / 2 pi ), v_rcp_f32, v_sqrt_f32, v_rsq_f32, and pow through v_log_f32 / v_exp_f32 in binary32;
// the translation unit is then also compiled with -fno-hip-fp32-correctly-rounded-divide-sqrt, so every `/` of the
// kernels is a reciprocal-based division.  acos and atan keep their polynomials (the reference calls the full-precision
// builtins there) over the native division and square root.  Images then agree with the exact mode statistically.
#pragma once

#include <hip/hip_runtime.h>

#include "pt_flavour.hpp"

namespace ptm {

#define PT_DEV __device__ __forceinline__

struct f3 { float x, y, z; };

PT_DEV f3 mk3( float x, float y, float z ) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 operator+( f3 a, f3 b ) { return mk3( a.x + b.x, a.y + b.y, a.z + b.z ); }
PT_DEV f3 operator-( f3 a, f3 b ) { return mk3( a.x - b.x, a.y - b.y, a.z - b.z ); }
PT_DEV f3 operator*( f3 a, f3 b ) { return mk3( a.x * b.x, a.y * b.y, a.z * b.z ); }
PT_DEV f3 operator*( f3 a, float s ) { return mk3( a.x * s, a.y * s, a.z * s ); }
PT_DEV f3 operator-( f3 a ) { return mk3( -a.x, -a.y, -a.z ); }
PT_DEV f3 yzx( f3 a ) { return mk3( a.y, a.z, a.x ); }

PT_DEV float fma1( float a, float b, float c ) { return __builtin_fmaf( a, b, c ); }
PT_DEV float fmin1( float a, float b ) { return __builtin_fminf( a, b ); }
PT_DEV float fmax1( float a, float b ) { return __builtin_fmaxf( a, b ); }
// a / b where the reference writes native_divide / native_recip or a plain `/` in shading code: the IEEE quotient, or
// (native) a * v_rcp_f32( b ) — two instructions where the compiler's 2.5-ulp division with denormal support takes eight
#if PT_ARITH_NATIVE
PT_DEV float div1( float a, float b ) { return a * __builtin_amdgcn_rcpf( b ); }
#else
PT_DEV float div1( float a, float b ) { return a / b; }
#endif
#if PT_ARITH_NATIVE
PT_DEV float sqrt1( float a ) { return __builtin_amdgcn_sqrtf( a ); }
PT_DEV float rsqrt1( float a ) { return __builtin_amdgcn_rsqf( a ); }
#else
PT_DEV float sqrt1( float a ) { return __builtin_sqrtf( a ); }
PT_DEV float rsqrt1( float a ) { return 1.0f / __builtin_sqrtf( a ); }
#endif
PT_DEV float inff() { return __builtin_inff(); }

// dot: z*z' + ( y*y' + x*x' ), two fmas
PT_DEV float dot( f3 a, f3 b ) {
	return fma1( a.z, b.z, fma1( a.y, b.y, a.x * b.x ) );
}

PT_DEV f3 cross( f3 a, f3 b ) {
	return mk3(
		fma1( a.y, b.z, -( a.z * b.y ) ),
		fma1( a.z, b.x, -( a.x * b.z ) ),
		fma1( a.x, b.y, -( a.y * b.x ) )
	);
}

PT_DEV f3 normalize( f3 a ) {
	const float inv = rsqrt1( dot( a, a ) );
	return a * inv;
}

// fma( s, a, b ) per component
PT_DEV f3 fma3( float s, f3 a, f3 b ) {
	return mk3( fma1( s, a.x, b.x ), fma1( s, a.y, b.y ), fma1( s, a.z, b.z ) );
}

// dir - 2 * dot( n, dir ) * n
PT_DEV f3 reflect( f3 dir, f3 n ) {
	const float s = 2.0f * dot( n, dir );
	return dir - n * s;
}

PT_DEV float max_cl( float x, float y ) { return ( x < y ) ? y : x; }
PT_DEV float clamp01( float x ) { return fmin1( fmax1( x, 0.0f ), 1.0f ); }

PT_DEV float fract( float x ) {
	return fmin1( x - __builtin_floorf( x ), 0x1.fffffep-1f );
}

// ---- sin / cos -------------------------------------------------------------------------
// k = rint( x * 2/pi ); r = x - k * pi/2 (three-part constant, fma); minimax on [-pi/4, pi/4].
PT_DEV void sincos( float x, float* sn, float* cs ) {
#if PT_ARITH_NATIVE
	// v_sin_f32 / v_cos_f32 take revolutions and are defined on [-256, 256]: reduce with v_fract_f32 first (the RNG's seed
	// grows by 1 per draw and by the hit distance per bounce, pt_utils.cl:39-44, pathtracing.cl:263)
	const float turns = __builtin_amdgcn_fractf( x * 0x1.45f306p-3f );
	*sn = __builtin_amdgcn_sinf( turns );
	*cs = __builtin_amdgcn_cosf( turns );
	return;
#endif
	if( !( __builtin_fabsf( x ) <= 1.0e8f ) ) {
		x = x * 0.0f;
	}

	const float k = __builtin_rintf( x * 0x1.45f306p-1f );
	float r = fma1( -k, 0x1.921fb6p+0f, x );
	r = fma1( -k, -0x1.777a5cp-25f, r );
	r = fma1( -k, -0x1.ee59dap-50f, r );
	const float z = r * r;

	float ps = fma1( -1.9515295891e-4f, z, 8.3321608736e-3f );
	ps = fma1( ps, z, -1.6666654611e-1f );
	const float s = fma1( ps * z, r, r );

	float pc = fma1( 2.443315711809948e-5f, z, -1.388731625493765e-3f );
	pc = fma1( pc, z, 4.166664568298827e-2f );
	const float c = fma1( pc * z, z, fma1( -0.5f, z, 1.0f ) );

	const int q = (int) ( k - 4.0f * __builtin_floorf( k * 0.25f ) );
	const bool swap = ( q & 1 ) != 0;
	const float a = swap ? c : s;   // |sin|
	const float b = swap ? s : c;   // |cos|
	*sn = ( q & 2 ) ? -a : a;
	*cs = ( q == 1 || q == 2 ) ? -b : b;
}

PT_DEV float sin1( float x ) { float s, c; sincos( x, &s, &c ); return s; }
PT_DEV float tan1( float x ) { float s, c; sincos( x, &s, &c ); return div1( s, c ); }

// ---- acos / atan -----------------------------------------------------------------------
PT_DEV float asin_core( float x ) {
	const float z = x * x;
	float p = fma1( 4.2163199048e-2f, z, 2.4181311049e-2f );
	p = fma1( p, z, 4.5470025998e-2f );
	p = fma1( p, z, 7.4953002686e-2f );
	p = fma1( p, z, 1.6666752422e-1f );
	return fma1( p * z, x, x );
}

PT_DEV float acos1( float x ) {
#if PT_ARITH_NATIVE
	// a quotient that is <= 1 in exact arithmetic (pt_brdf.cl:197: a / ( rough - a * rough + a )) can come out an ulp above it
	// from a reciprocal-based division, and acos of that is NaN: one v_med3_f32
	x = __builtin_amdgcn_fmed3f( x, -1.0f, 1.0f );
#endif
	if( x < -0.5f ) {
		return 0x1.921fb6p+1f - 2.0f * asin_core( sqrt1( 0.5f * ( 1.0f + x ) ) );
	}
	if( x > 0.5f ) {
		return 2.0f * asin_core( sqrt1( 0.5f * ( 1.0f - x ) ) );
	}
	return ( 0x1.921fb6p+0f - asin_core( x ) ) + -0x1.777a5cp-25f;
}

PT_DEV float atan1( float xx ) {
	const float ax = __builtin_fabsf( xx );
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
	float p = fma1( 8.05374449538e-2f, z, -1.38776856032e-1f );
	p = fma1( p, z, 1.99777106478e-1f );
	p = fma1( p, z, -3.33329491539e-1f );
	const float y = y0 + fma1( p * z, x, x );

	return __builtin_copysignf( y, xx );
}

// ---- pow: exp2( y * log2( x ) ) in binary64 ----------------------------------------------
PT_DEV double fmad( double a, double b, double c ) { return __builtin_fma( a, b, c ); }

// fma( a, b, c ) with the constant c in a scalar register pair.  A binary64 literal cannot be an operand, and left to
// itself the compiler puts each polynomial coefficient into a VGPR pair with two v_mov_b32 next to every v_fma_f64: a
// third of pow's vector instructions.  As an SGPR operand the constant costs two s_mov_b32 on the scalar unit, which
// these kernels leave half idle, and nothing on the vector unit that bounds them.  Same operation, same result.
PT_DEV double fmaConst( double a, double b, double c ) {
	double r;
	asm( "v_fma_f64 %0, %1, %2, %3" : "=v"( r ) : "v"( a ), "v"( b ), "s"( c ) );
	return r;
}

PT_DEV double log2_d( double a ) {
	const unsigned long long bits = (unsigned long long) __double_as_longlong( a );
	int e = (int) ( ( bits >> 52 ) & 0x7ffULL ) - 1023;
	double m = __longlong_as_double( (long long) ( ( bits & 0x000fffffffffffffULL ) | 0x3ff0000000000000ULL ) );

	if( m > 0x1.6a09e667f3bcdp+0 ) {
		m *= 0.5;
		e += 1;
	}

	const double s = ( m - 1.0 ) / ( m + 1.0 );
	const double s2 = s * s;
	double p = 0x1.e1e1e1e1e1e1ep-5;
	p = fmaConst( p, s2, 0x1.1111111111111p-4 );
	p = fmaConst( p, s2, 0x1.3b13b13b13b14p-4 );
	p = fmaConst( p, s2, 0x1.745d1745d1746p-4 );
	p = fmaConst( p, s2, 0x1.c71c71c71c71cp-4 );
	p = fmaConst( p, s2, 0x1.2492492492492p-3 );
	p = fmaConst( p, s2, 0x1.999999999999ap-3 );
	p = fmaConst( p, s2, 0x1.5555555555555p-2 );
	p = fmad( p, s2, 1.0 );
	const double ln_m = 2.0 * s * p;

	return fmad( ln_m, 0x1.71547652b82fep+0, (double) e );
}

PT_DEV double exp2_d( double t ) {
	const double n = __builtin_rint( t );
	const double g = ( t - n ) * 0x1.62e42fefa39efp-1;
	double p = 0x1.6124613a86d09p-33;
	p = fmaConst( p, g, 0x1.1eed8eff8d898p-29 );
	p = fmaConst( p, g, 0x1.ae64567f544e4p-26 );
	p = fmaConst( p, g, 0x1.27e4fb7789f5cp-22 );
	p = fmaConst( p, g, 0x1.71de3a556c734p-19 );
	p = fmaConst( p, g, 0x1.a01a01a01a01ap-16 );
	p = fmaConst( p, g, 0x1.a01a01a01a01ap-13 );
	p = fmaConst( p, g, 0x1.6c16c16c16c17p-10 );
	p = fmaConst( p, g, 0x1.1111111111111p-7 );
	p = fmaConst( p, g, 0x1.5555555555555p-5 );
	p = fmaConst( p, g, 0x1.5555555555555p-3 );
	p = fmad( p, g, 0.5 );
	p = fmad( p, g, 1.0 );
	p = fmad( p, g, 1.0 );
	const long long biased = (long long) n + 1023;
	const double scale = __longlong_as_double( biased << 52 );

	return p * scale;
}

PT_DEV float pow1( float x, float y ) {
	if( y == 0.0f || x == 1.0f ) {
		return 1.0f;
	}
	if( x != x || y != y ) {
		return x + y;
	}

	const float ay = __builtin_fabsf( y );
	const float ax = __builtin_fabsf( x );
	const bool yInt = ( ay >= 0x1p24f ) || ( __builtin_floorf( ay ) == ay );
	const float half = ay * 0.5f;
	const bool yOdd = yInt && ( ay < 0x1p24f ) && ( __builtin_floorf( half ) != half );
	const bool xNeg = ( __float_as_uint( x ) >> 31 ) != 0u;
	float sign = 1.0f;

	if( xNeg ) {
		if( yOdd ) {
			sign = -1.0f;
		}
		else if( !yInt && ax != 0.0f && ax != inff() ) {
			return inff() - inff();
		}
	}

	if( ax == 1.0f ) {
		return sign;
	}

#if PT_ARITH_NATIVE
	// v_log_f32 / v_exp_f32 ( log2( 0 ) = -inf, exp2( -inf ) = 0, log2( inf ) = inf )
	return sign * __builtin_amdgcn_exp2f( y * __builtin_amdgcn_logf( ax ) );
#endif
	double l;

	if( ax == 0.0f ) {
		l = -(double) inff();
	}
	else if( ax == inff() ) {
		l = (double) inff();
	}
	else {
		l = log2_d( (double) ax );
	}

	double t = (double) y * l;
	t = ( t > 130.0 ) ? 130.0 : t;
	t = ( t < -160.0 ) ? -160.0 : t;

	return sign * (float) exp2_d( t );
}

// cbrt (solveCubic, pt_utils.cl:153): through the binary64 log2 / exp2 of pow1
PT_DEV float cbrt1( float x ) {
	const float ax = __builtin_fabsf( x );

	if( x != x || ax == 0.0f || ax == inff() ) {
		return x;
	}

	return __builtin_copysignf( (float) exp2_d( log2_d( (double) ax ) / 3.0 ), x );
}

PT_DEV float cos1( float x ) { float s, c; sincos( x, &s, &c ); return c; }

// pow1 as a real function call.  Inlined, its binary64 polynomial chains get interleaved with the caller's
// shading code and raise its register pressure; kernels with the 64-register budget (and the lane state machine)
// call it instead — 128 -> 240 B less scratch per lane, +3..6 % on the 260k - 2M triangle scenes, -1 % for the
// lean lock-step kernel, which therefore keeps the inlined form (template flag CALLS in pt_kernel.hpp).
__device__ __attribute__( ( noinline ) ) float pow1Call( float x, float y ) {
	return pow1( x, y );
}

template<bool CALLS>
PT_DEV float powSelect( float x, float y ) {
	return CALLS ? pow1Call( x, y ) : pow1( x, y );
}

}  // namespace ptm
