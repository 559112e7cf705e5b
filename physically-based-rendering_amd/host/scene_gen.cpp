#include "scene_gen.h"

#include <cmath>
#include <stdexcept>

namespace pbr {

namespace {

// ---- reproducible numerics: nothing here calls libm's transcendental functions ----

struct Rng {
	uint64_t s;

	explicit Rng( uint64_t seed ) : s( seed * 0x9e3779b97f4a7c15ULL + 0x632be59bd9b4e019ULL ) {}

	uint64_t next() {  // splitmix64
		uint64_t z = ( s += 0x9e3779b97f4a7c15ULL );
		z = ( z ^ ( z >> 30 ) ) * 0xbf58476d1ce4e5b9ULL;
		z = ( z ^ ( z >> 27 ) ) * 0x94d049bb133111ebULL;
		return z ^ ( z >> 31 );
	}

	double uniform() { return (double) ( next() >> 11 ) * ( 1.0 / 9007199254740992.0 ); }
	double range( double a, double b ) { return a + ( b - a ) * uniform(); }
};

const double kPi = 3.14159265358979323846;

// sin / cos by quadrant reduction + Taylor series (double; |error| < 1e-15 for |x| < 1e4)
void sincosd( double x, double* s, double* c ) {
	const double k = std::floor( x * ( 2.0 / kPi ) + 0.5 );
	const double r = ( x - k * 1.5707963267948966 ) - k * 6.123233995736766e-17;
	const double z = r * r;
	double ps = 1.0, pc = 1.0, ts = 1.0, tc = 1.0;

	for( int i = 1; i <= 10; i++ ) {
		tc *= -z / (double) ( ( 2 * i - 1 ) * ( 2 * i ) );
		ts *= -z / (double) ( ( 2 * i ) * ( 2 * i + 1 ) );
		pc += tc;
		ps += ts;
	}

	const double sr = r * ps, cr = pc;
	const int q = (int) ( k - 4.0 * std::floor( k * 0.25 ) );
	*s = ( q == 0 ) ? sr : ( q == 1 ) ? cr : ( q == 2 ) ? -sr : -cr;
	*c = ( q == 0 ) ? cr : ( q == 1 ) ? -sr : ( q == 2 ) ? -cr : sr;
}

double sind( double x ) { double s, c; sincosd( x, &s, &c ); return s; }
double cosd( double x ) { double s, c; sincosd( x, &s, &c ); return c; }

struct V { double x, y, z; };

V operator+( V a, V b ) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
V operator-( V a, V b ) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
V operator*( V a, double s ) { return { a.x * s, a.y * s, a.z * s }; }
V cross( V a, V b ) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
double dot( V a, V b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }
V unit( V a ) { const double l = std::sqrt( dot( a, a ) ); return ( l > 0.0 ) ? a * ( 1.0 / l ) : V{ 0.0, 1.0, 0.0 }; }


// ---- mesh assembly in ObjParser's output form ----

struct Builder {
	SceneModel* m;

	explicit Builder( SceneModel* model ) : m( model ) {
		// one shared normal: the path reads vertex normals only under Phong tessellation
		m->normals.push_back( 0.0f );
		m->normals.push_back( 1.0f );
		m->normals.push_back( 0.0f );
	}

	int material( const material_t& mtl ) {
		m->materials.push_back( mtl );
		return (int) m->materials.size() - 1;
	}

	void object( const std::string& name ) {
		object3D o;
		o.oName = name;
		m->objects.push_back( o );
	}

	uint32_t vertex( V p ) {
		m->vertices.push_back( (float) p.x );
		m->vertices.push_back( (float) p.y );
		m->vertices.push_back( (float) p.z );
		return (uint32_t) ( m->vertices.size() / 3 - 1 );
	}

	void tri( uint32_t a, uint32_t b, uint32_t c, int mtl ) {
		const uint32_t v[3] = { a, b, c };
		object3D& o = m->objects.back();

		for( int k = 0; k < 3; k++ ) {
			m->facesV.push_back( v[k] );
			m->facesVN.push_back( 0 );
			o.facesV.push_back( v[k] );
			o.facesVN.push_back( 0 );
		}

		m->facesMtl.push_back( mtl );
	}

	void quad( uint32_t a, uint32_t b, uint32_t c, uint32_t d, int mtl ) {
		tri( a, b, c, mtl );
		tri( a, c, d, mtl );
	}

	void quad( V a, V b, V c, V d, int mtl ) {
		quad( vertex( a ), vertex( b ), vertex( c ), vertex( d ), mtl );
	}

	// parallelogram p + s*eu + t*ev tessellated nu x nv
	void grid( V p, V eu, V ev, int nu, int nv, int mtl ) {
		std::vector<uint32_t> idx( (size_t) ( nu + 1 ) * ( nv + 1 ) );

		for( int j = 0; j <= nv; j++ ) {
			for( int i = 0; i <= nu; i++ ) {
				idx[(size_t) j * ( nu + 1 ) + i] = vertex( p + eu * ( (double) i / nu ) + ev * ( (double) j / nv ) );
			}
		}

		for( int j = 0; j < nv; j++ ) {
			for( int i = 0; i < nu; i++ ) {
				const size_t a = (size_t) j * ( nu + 1 ) + i;
				quad( idx[a], idx[a + 1], idx[a + nu + 2], idx[a + nu + 1], mtl );
			}
		}
	}

	// cuboid: centre of the base c, half sizes hx / hz, height h, rotation (cs, sn) about y; 12 tris
	void block( V c, double hx, double hz, double h, double cs, double sn, int mtl, bool withBottom = true ) {
		uint32_t v[8];

		for( int k = 0; k < 8; k++ ) {
			const double lx = ( k & 1 ) ? hx : -hx;
			const double lz = ( k & 2 ) ? hz : -hz;
			const double ly = ( k & 4 ) ? h : 0.0;
			v[k] = vertex( { c.x + lx * cs + lz * sn, c.y + ly, c.z - lx * sn + lz * cs } );
		}

		quad( v[4], v[5], v[7], v[6], mtl );  // top
		quad( v[0], v[1], v[5], v[4], mtl );
		quad( v[1], v[3], v[7], v[5], mtl );
		quad( v[3], v[2], v[6], v[7], mtl );
		quad( v[2], v[0], v[4], v[6], mtl );

		if( withBottom ) {
			quad( v[0], v[2], v[3], v[1], mtl );
		}
	}

	// closed-in-u surface of revolution style grid: ring(i) of nv points, wrapped in v
	void tube( const std::vector<std::vector<V>>& rings, bool closeU, int mtl ) {
		const size_t nu = rings.size();
		const size_t nv = rings[0].size();
		std::vector<uint32_t> idx( nu * nv );

		for( size_t i = 0; i < nu; i++ ) {
			for( size_t j = 0; j < nv; j++ ) {
				idx[i * nv + j] = vertex( rings[i][j] );
			}
		}

		const size_t lastU = closeU ? nu : nu - 1;

		for( size_t i = 0; i < lastU; i++ ) {
			const size_t i1 = ( i + 1 ) % nu;

			for( size_t j = 0; j < nv; j++ ) {
				const size_t j1 = ( j + 1 ) % nv;
				quad( idx[i * nv + j], idx[i1 * nv + j], idx[i1 * nv + j1], idx[i * nv + j1], mtl );
			}
		}
	}
};

material_t mtlDiffuse( const char* name, float r, float g, float b ) {
	material_t m;
	m.mtlName = name;
	m.Kd = { r, g, b, 0.0f };
	return m;
}

struct Palette { int white, red, green, glossy, glass, stone, cloth, sky; };

Palette addPalette( Builder* b ) {
	Palette p;
	p.white = b->material( mtlDiffuse( "White", 0.73f, 0.73f, 0.73f ) );
	p.red = b->material( mtlDiffuse( "Red", 0.65f, 0.05f, 0.05f ) );
	p.green = b->material( mtlDiffuse( "Green", 0.12f, 0.45f, 0.15f ) );

	material_t glossy = mtlDiffuse( "Glossy", 0.8f, 0.8f, 0.85f );
	glossy.Ks = { 0.9f, 0.9f, 0.9f, 0.0f };
	glossy.nu = 200.0f;
	glossy.nv = 200.0f;
	glossy.Rs = 0.7f;
	glossy.Rd = 0.6f;
	glossy.rough = 0.15f;
	p.glossy = b->material( glossy );

	material_t glass = mtlDiffuse( "Glass", 0.95f, 0.95f, 1.0f );
	glass.d = 0.1f;
	glass.Ni = 1.5f;
	glass.Rs = 0.2f;
	glass.rough = 0.05f;
	p.glass = b->material( glass );

	material_t stone = mtlDiffuse( "Stone", 0.62f, 0.58f, 0.5f );
	stone.nu = 8.0f;
	stone.nv = 8.0f;
	stone.Rs = 0.08f;
	stone.rough = 0.8f;
	p.stone = b->material( stone );

	p.cloth = b->material( mtlDiffuse( "Cloth", 0.55f, 0.12f, 0.1f ) );
	p.sky = b->material( mtlDiffuse( "sky_light", 0.846f, 0.933f, 0.949f ) );
	return p;
}

// The open-fronted, open-topped room the small scenes sit in: x in [-1,1], y in [0,2], z in [-1,1]
void addRoom( Builder* b, const Palette& p, bool ceilingWithHole ) {
	b->object( "Walls" );
	b->quad( V{ -1, 0, 1 }, V{ 1, 0, 1 }, V{ 1, 0, -1 }, V{ -1, 0, -1 }, p.white );   // floor
	b->quad( V{ -1, 0, -1 }, V{ 1, 0, -1 }, V{ 1, 2, -1 }, V{ -1, 2, -1 }, p.white );  // back
	b->quad( V{ -1, 0, 1 }, V{ -1, 0, -1 }, V{ -1, 2, -1 }, V{ -1, 2, 1 }, p.red );    // left
	b->quad( V{ 1, 0, -1 }, V{ 1, 0, 1 }, V{ 1, 2, 1 }, V{ 1, 2, -1 }, p.green );      // right

	if( ceilingWithHole ) {
		const double h = 0.5;
		b->quad( V{ -1, 2, -1 }, V{ 1, 2, -1 }, V{ 1, 2, -h }, V{ -1, 2, -h }, p.white );
		b->quad( V{ -1, 2, h }, V{ 1, 2, h }, V{ 1, 2, 1 }, V{ -1, 2, 1 }, p.white );
		b->quad( V{ -1, 2, -h }, V{ -h, 2, -h }, V{ -h, 2, h }, V{ -1, 2, h }, p.white );
		b->quad( V{ h, 2, -h }, V{ 1, 2, -h }, V{ 1, 2, h }, V{ h, 2, h }, p.white );
	}
}


// ---- config 1 / 2: Cornell-class box ----

void genCornell( GeneratedScene* out ) {
	Builder b( &out->model );
	const Palette p = addPalette( &b );

	addRoom( &b, p, true );

	b.object( "ShortBlock" );
	b.block( V{ 0.35, 0.0, 0.35 }, 0.3, 0.3, 0.6, 0.96, 0.28, p.white );

	b.object( "TallBlock" );
	b.block( V{ -0.35, 0.0, -0.3 }, 0.3, 0.3, 1.2, 0.96, -0.28, p.glossy );

	// glass octahedron standing on the short block
	b.object( "Crystal" );
	{
		const V c = { 0.35, 0.85, 0.35 };
		const double r = 0.2;
		const uint32_t top = b.vertex( { c.x, c.y + r * 1.2, c.z } );
		const uint32_t bot = b.vertex( { c.x, c.y - r * 1.2, c.z } );
		const uint32_t e[4] = {
			b.vertex( { c.x + r, c.y, c.z } ), b.vertex( { c.x, c.y, c.z + r } ),
			b.vertex( { c.x - r, c.y, c.z } ), b.vertex( { c.x, c.y, c.z - r } )
		};

		for( int k = 0; k < 4; k++ ) {
			b.tri( e[k], e[( k + 1 ) & 3], top, p.glass );
			b.tri( e[( k + 1 ) & 3], e[k], bot, p.glass );
		}
	}

	out->eye[0] = 0.0f; out->eye[1] = 1.0f; out->eye[2] = 3.0f;
	out->center[0] = 0.0f; out->center[1] = 0.0f; out->center[2] = 1.0f;
}


// ---- config 3: Dragon-class — one closed, displaced (2,3) torus-knot tube in the open room ----

void genDragon( GeneratedScene* out, uint32_t seed, uint32_t triangles ) {
	Builder b( &out->model );
	const Palette p = addPalette( &b );
	Rng rng( seed );

	addRoom( &b, p, false );

	const int nv = 200;
	int nu = (int) ( ( triangles > 64 ? triangles - 10 : 64 ) / ( 2 * nv ) );
	nu = ( nu < 16 ) ? 16 : nu;

	double phase[6];

	for( int k = 0; k < 6; k++ ) {
		phase[k] = rng.range( 0.0, 2.0 * kPi );
	}

	const double scale = 0.26;
	const V centre = { 0.0, 1.0, 0.0 };
	std::vector<std::vector<V>> rings( (size_t) nu, std::vector<V>( (size_t) nv ) );

	for( int i = 0; i < nu; i++ ) {
		const double t = 2.0 * kPi * (double) i / nu;
		double s2, c2, s3, c3;
		sincosd( 2.0 * t, &s2, &c2 );
		sincosd( 3.0 * t, &s3, &c3 );

		const V pos = { ( 2.0 + c3 ) * c2, s3, ( 2.0 + c3 ) * s2 };
		// analytic tangent
		const V tan = unit( { -3.0 * s3 * c2 - 2.0 * ( 2.0 + c3 ) * s2, 3.0 * c3, -3.0 * s3 * s2 + 2.0 * ( 2.0 + c3 ) * c2 } );
		// frame from the direction away from the knot's axis (never parallel to the tangent)
		const V radial = unit( { c2, 0.0, s2 } );
		const V nrm = unit( radial - tan * dot( radial, tan ) );
		const V bin = cross( tan, nrm );

		for( int j = 0; j < nv; j++ ) {
			const double a = 2.0 * kPi * (double) j / nv;
			double sa, ca;
			sincosd( a, &sa, &ca );

			const double bump =
				0.16 * sind( 23.0 * t + 3.0 * a + phase[0] ) +
				0.10 * sind( 61.0 * t - 5.0 * a + phase[1] ) +
				0.06 * sind( 149.0 * t + 11.0 * a + phase[2] ) +
				0.04 * sind( 7.0 * a + phase[3] ) * cosd( 311.0 * t + phase[4] ) +
				0.03 * sind( 467.0 * t + 17.0 * a + phase[5] );
			const double r = 0.55 * ( 1.0 + bump );
			const V q = pos + nrm * ( r * ca ) + bin * ( r * sa );

			rings[(size_t) i][(size_t) j] = centre + q * scale;
		}
	}

	b.object( "Knot" );
	b.tube( rings, true, p.glossy );

	out->eye[0] = 0.0f; out->eye[1] = 1.0f; out->eye[2] = 3.0f;
	out->center[0] = 0.0f; out->center[1] = 0.0f; out->center[2] = 1.0f;
}


// ---- config 4: Sponza-class — colonnaded two-storey atrium, many objects, open roof ----

void genSponza( GeneratedScene* out, uint32_t seed, uint32_t triangles ) {
	Builder b( &out->model );
	const Palette p = addPalette( &b );
	Rng rng( seed );

	// tessellation scale: the counts below give ~226k triangles at s = 1
	const double s = std::sqrt( (double) ( triangles > 2000 ? triangles : 2000 ) / 226000.0 );
	auto seg = [s]( int n, int lo ) {
		const int v = (int) ( n * s + 0.5 );
		return ( v < lo ) ? lo : v;
	};

	const double X = 6.0, Z = 3.0, H = 6.0;

	b.object( "Floor" );
	b.grid( V{ -X, 0, Z }, V{ 2 * X, 0, 0 }, V{ 0, 0, -2 * Z }, seg( 60, 2 ), seg( 30, 2 ), p.stone );

	const char* wallNames[4] = { "WallNorth", "WallSouth", "WallWest", "WallEast" };
	b.object( wallNames[0] );
	b.grid( V{ -X, 0, -Z }, V{ 2 * X, 0, 0 }, V{ 0, H, 0 }, seg( 40, 2 ), seg( 20, 2 ), p.white );
	b.object( wallNames[1] );
	b.grid( V{ X, 0, Z }, V{ -2 * X, 0, 0 }, V{ 0, H, 0 }, seg( 40, 2 ), seg( 20, 2 ), p.white );
	b.object( wallNames[2] );
	b.grid( V{ -X, 0, Z }, V{ 0, 0, -2 * Z }, V{ 0, H, 0 }, seg( 40, 2 ), seg( 20, 2 ), p.red );
	b.object( wallNames[3] );
	b.grid( V{ X, 0, -Z }, V{ 0, 0, 2 * Z }, V{ 0, H, 0 }, seg( 40, 2 ), seg( 20, 2 ), p.green );

	const int perRow = 10;
	const double rowZ[2] = { -1.8, 1.8 };
	const double storeyY[2] = { 0.0, 3.0 };
	const double colH = 2.6, colR = 0.17;
	const int cs = seg( 32, 6 ), cr = seg( 40, 2 );

	for( int storey = 0; storey < 2; storey++ ) {
		for( int row = 0; row < 2; row++ ) {
			for( int k = 0; k < perRow; k++ ) {
				const double cx = -X + 0.6 + ( 2 * X - 1.2 ) * (double) k / ( perRow - 1 );
				const V base = { cx, storeyY[storey], rowZ[row] };
				const double flute = rng.range( 0.0, 2.0 * kPi );

				b.object( "Column" + std::to_string( storey ) + "_" + std::to_string( row ) + "_" + std::to_string( k ) );
				b.block( base, 0.24, 0.24, 0.15, 1.0, 0.0, p.stone );
				b.block( V{ base.x, base.y + colH - 0.15, base.z }, 0.24, 0.24, 0.15, 1.0, 0.0, p.stone );

				std::vector<std::vector<V>> rings( (size_t) cr + 1, std::vector<V>( (size_t) cs ) );

				for( int i = 0; i <= cr; i++ ) {
					const double y = 0.15 + ( colH - 0.3 ) * (double) i / cr;
					const double taper = 1.0 - 0.12 * (double) i / cr;

					for( int j = 0; j < cs; j++ ) {
						const double a = 2.0 * kPi * (double) j / cs;
						double sa, ca;
						sincosd( a, &sa, &ca );
						const double r = colR * taper * ( 1.0 + 0.04 * cosd( 12.0 * a + flute ) );
						rings[(size_t) i][(size_t) j] = { base.x + r * ca, base.y + y, base.z + r * sa };
					}
				}

				b.tube( rings, false, ( storey == 0 ) ? p.stone : p.white );
			}
		}
	}

	// arches between neighbouring ground-floor columns
	const int au = seg( 48, 4 ), av = seg( 24, 4 );

	for( int row = 0; row < 2; row++ ) {
		for( int k = 0; k + 1 < perRow; k++ ) {
			const double x0 = -X + 0.6 + ( 2 * X - 1.2 ) * (double) k / ( perRow - 1 );
			const double x1 = -X + 0.6 + ( 2 * X - 1.2 ) * (double) ( k + 1 ) / ( perRow - 1 );
			const double mid = 0.5 * ( x0 + x1 ), rad = 0.5 * ( x1 - x0 );
			std::vector<std::vector<V>> rings( (size_t) au + 1, std::vector<V>( (size_t) av ) );

			for( int i = 0; i <= au; i++ ) {
				const double t = kPi * (double) i / au;
				double st, ct;
				sincosd( t, &st, &ct );
				const V c = { mid - rad * ct, colH + 0.05 + 0.35 * st, rowZ[row] };
				const V nrm = { -ct, st * 0.35 / rad, 0.0 };
				const V n1 = unit( nrm );

				for( int j = 0; j < av; j++ ) {
					const double a = 2.0 * kPi * (double) j / av;
					double sa, ca;
					sincosd( a, &sa, &ca );
					rings[(size_t) i][(size_t) j] = c + n1 * ( 0.09 * ca ) + V{ 0.0, 0.0, 1.0 } * ( 0.16 * sa );
				}
			}

			b.object( "Arch" + std::to_string( row ) + "_" + std::to_string( k ) );
			b.tube( rings, false, p.stone );
		}
	}

	// gallery slabs between the colonnades and the side walls, and roof beams
	b.object( "GalleryNorth" );
	b.block( V{ 0.0, 2.75, -2.4 }, X, 0.6, 0.25, 1.0, 0.0, p.white );
	b.object( "GallerySouth" );
	b.block( V{ 0.0, 2.75, 2.4 }, X, 0.6, 0.25, 1.0, 0.0, p.white );

	for( int k = 0; k < 5; k++ ) {
		b.object( "Beam" + std::to_string( k ) );
		b.block( V{ -4.8 + 2.4 * k, H - 0.3, 0.0 }, 0.12, Z, 0.3, 1.0, 0.0, p.stone );
	}

	// hanging drapes with sine folds
	const int du = seg( 60, 4 ), dv = seg( 100, 4 );

	for( int k = 0; k < 6; k++ ) {
		const double cx = -4.5 + 1.8 * k;
		const double z0 = ( k & 1 ) ? 1.15 : -1.15;
		const double ph = rng.range( 0.0, 2.0 * kPi );
		std::vector<uint32_t> idx( (size_t) ( du + 1 ) * ( dv + 1 ) );

		b.object( "Drape" + std::to_string( k ) );

		for( int j = 0; j <= dv; j++ ) {
			for( int i = 0; i <= du; i++ ) {
				const double u = (double) i / du, v = (double) j / dv;
				const double fold = 0.07 * ( 0.3 + v ) * sind( 28.0 * u + ph + 3.0 * v );
				idx[(size_t) j * ( du + 1 ) + i] = b.vertex( { cx - 0.55 + 1.1 * u, 5.2 - 2.4 * v, z0 + fold } );
			}
		}

		for( int j = 0; j < dv; j++ ) {
			for( int i = 0; i < du; i++ ) {
				const size_t a = (size_t) j * ( du + 1 ) + i;
				b.quad( idx[a], idx[a + 1], idx[a + du + 2], idx[a + du + 1], ( k % 3 == 0 ) ? p.cloth : ( k % 3 == 1 ) ? p.green : p.glossy );
			}
		}
	}

	out->eye[0] = -5.3f; out->eye[1] = 1.7f; out->eye[2] = 0.15f;
	out->center[0] = 1.0f; out->center[1] = -0.12f; out->center[2] = 0.02f;
}


// ---- config 5: hairball — thin random-walk triangle strips inside a ball ----

void genHairball( GeneratedScene* out, uint32_t seed, uint32_t triangles ) {
	Builder b( &out->model );
	const Palette p = addPalette( &b );
	Rng rng( seed );

	b.object( "Ground" );
	b.quad( V{ -3, -0.05, 3 }, V{ 3, -0.05, 3 }, V{ 3, -0.05, -3 }, V{ -3, -0.05, -3 }, p.white );

	const int segs = 500;
	int strands = (int) ( ( triangles > 4 ? triangles - 2 : 4 ) / ( 2 * segs ) );
	strands = ( strands < 1 ) ? 1 : strands;

	const V centre = { 0.0, 1.0, 0.0 };
	const double R = 1.0, step = 0.02, halfWidth = 0.003;

	b.object( "Hair" );

	for( int sIdx = 0; sIdx < strands; sIdx++ ) {
		// start on a random point of a small inner sphere, heading outwards
		V d = unit( { rng.range( -1, 1 ), rng.range( -1, 1 ), rng.range( -1, 1 ) } );
		V pos = d * 0.15;
		const int mtl = ( sIdx % 7 == 0 ) ? p.glossy : ( sIdx % 3 == 0 ) ? p.cloth : p.stone;
		uint32_t prevA = 0, prevB = 0;

		for( int k = 0; k <= segs; k++ ) {
			const V side = unit( cross( d, ( std::fabs( d.y ) < 0.9 ) ? V{ 0, 1, 0 } : V{ 1, 0, 0 } ) );
			const uint32_t a = b.vertex( centre + pos + side * halfWidth );
			const uint32_t c = b.vertex( centre + pos - side * halfWidth );

			if( k > 0 ) {
				b.quad( prevA, prevB, c, a, mtl );
			}

			prevA = a;
			prevB = c;

			// persistent random walk, reflected at the ball's surface
			d = unit( d + V{ rng.range( -1, 1 ), rng.range( -1, 1 ), rng.range( -1, 1 ) } * 0.35 );
			V next = pos + d * step;

			if( dot( next, next ) > R * R ) {
				const V n = unit( pos );
				d = unit( d - n * ( 2.0 * dot( d, n ) ) );
				next = pos + d * step;
			}

			pos = next;
		}
	}

	out->eye[0] = 0.0f; out->eye[1] = 1.0f; out->eye[2] = 3.0f;
	out->center[0] = 0.0f; out->center[1] = 0.0f; out->center[2] = 1.0f;
}

}  // namespace


GeneratedScene generateScene( const std::string& kind, uint32_t seed, uint32_t triangles ) {
	GeneratedScene out;

	if( kind == "cornell" ) {
		genCornell( &out );
	}
	else if( kind == "dragon" ) {
		genDragon( &out, seed, triangles ? triangles : 870000u );
	}
	else if( kind == "sponza" ) {
		genSponza( &out, seed, triangles ? triangles : 260000u );
	}
	else if( kind == "hairball" ) {
		genHairball( &out, seed, triangles ? triangles : 2000000u );
	}
	else {
		throw std::runtime_error( "unknown scene kind \"" + kind + "\"" );
	}

	return out;
}

}  // namespace pbr
