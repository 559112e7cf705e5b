// extern "C" surface of libpbrhost.so for the Python test / bench harness (ctypes).
// It exposes the host half of the path — config, scene load / generation, BVH build,
// buffer packing, camera, PathTracer driver — with plain pointers, so the harness can hand the
// SAME flat arrays to the HIP core (libpbrhip.so) and to the CPU oracle.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "Cfg.h"
#include "bvh_builder.h"
#include "model_io.h"
#include "cl_adaptor.h"
#include "path_tracer.h"
#include "scene_gen.h"

using namespace pbr;

namespace {

struct HostScene {
	ModelLoader loader;
	std::unique_ptr<BVH> bvh;
	SceneBuffers buffers;
	float eye[3] = { 0.0f, 1.0f, 3.0f };
	float center[3] = { 0.0f, 0.0f, 1.0f };
	bool hasCamera = false;
};

thread_local std::string gError;

void finishScene( HostScene* s ) {
	const SceneModel& m = s->loader.getObjParser()->model();
	s->bvh.reset( new BVH( m.objects, m.vertices, m.normals ) );
	s->buffers.build( &s->loader, s->bvh.get() );
}

}  // namespace

extern "C" {

const char* pbrh_last_error() {
	return gError.c_str();
}

// ---- Cfg -------------------------------------------------------------------------------

void pbrh_cfg_reset() {
	Cfg::get().resetDefaults();
}

void pbrh_cfg_set( const char* key, const char* value ) {
	Cfg::get().value<std::string>( key, std::string( value ) );
}

int pbrh_cfg_get( const char* key, char* out, int len ) {
	const std::string v = Cfg::get().value<std::string>( key );
	snprintf( out, (size_t) len, "%s", v.c_str() );
	return (int) v.size();
}

int pbrh_cfg_load( const char* path ) {
	return Cfg::get().loadConfigFile( path ) ? 0 : -1;
}

// ---- scenes ----------------------------------------------------------------------------

// ModelLoader::loadModel + new BVH + the buffer packing of PathTracer::initOpenCLBuffers,
// i.e. GLWidget::loadModel (source/qt/GLWidget.cpp:339-387) without the GL parts.
void* pbrh_scene_load_obj( const char* dir, const char* file ) {
	try {
		std::unique_ptr<HostScene> s( new HostScene() );
		s->loader.loadModel( dir, file );

		if( s->loader.getObjParser()->model().facesV.empty() ) {
			gError = std::string( "no faces loaded from " ) + dir + file;
			return nullptr;
		}

		finishScene( s.get() );
		return s.release();
	}
	catch( const std::exception& e ) {
		gError = e.what();
		return nullptr;
	}
}

void* pbrh_scene_generate( const char* kind, uint32_t seed, uint32_t triangles ) {
	try {
		std::unique_ptr<HostScene> s( new HostScene() );
		GeneratedScene g = generateScene( kind, seed, triangles );
		s->loader.getObjParser()->adopt( g.model );
		std::memcpy( s->eye, g.eye, sizeof( g.eye ) );
		std::memcpy( s->center, g.center, sizeof( g.center ) );
		s->hasCamera = true;
		finishScene( s.get() );
		return s.release();
	}
	catch( const std::exception& e ) {
		gError = e.what();
		return nullptr;
	}
}

void pbrh_scene_destroy( void* scene ) {
	delete static_cast<HostScene*>( scene );
}

void pbrh_scene_desc( void* scene, pbr_scene_desc* out ) {
	*out = static_cast<HostScene*>( scene )->buffers.desc();
}

// out[0..7] = flat nodes, faces, vertices, materials, lights, tree nodes before skip-ahead
// deletion, leaves, max depth; out[8] = skip-ahead marks; out[9] = objects
void pbrh_scene_info( void* scene, uint32_t* out ) {
	HostScene* s = static_cast<HostScene*>( scene );
	const pbr_scene_desc d = s->buffers.desc();
	out[0] = d.num_nodes;
	out[1] = d.num_faces;
	out[2] = d.num_vertices;
	out[3] = d.num_materials;
	out[4] = d.num_lights;
	out[5] = (uint32_t) s->bvh->nodes().size();
	out[6] = (uint32_t) s->bvh->getLeafNodes().size();
	out[7] = s->bvh->getDepth();
	out[8] = s->bvh->numSkipped();
	out[9] = (uint32_t) s->loader.getObjParser()->model().objects.size();
}

// scripts/bvh_sweep.py: builder variants (BVH::LAB_*), 0 = the reference's builder
void pbrh_lab_bvh_flags( unsigned flags ) {
	BVH::sLabFlags = flags;
}

void pbrh_scene_config( void* scene, uint32_t width, uint32_t height, pbr_config* out ) {
	*out = PathTracer::makeConfig( static_cast<HostScene*>( scene )->buffers, width, height );
}

// Camera for the scene: the generator's suggestion, else the Cfg default pose
// (config.json:3-18); focal length / aperture from Cfg; no focus point.
void pbrh_scene_camera( void* scene, pbr_camera* out ) {
	HostScene* s = static_cast<HostScene*>( scene );
	Camera cam;

	if( s->hasCamera ) {
		cam.setEye( s->eye[0], s->eye[1], s->eye[2] );
		cam.setCenter( s->center[0], s->center[1], s->center[2] );
	}

	*out = pbr_camera();
	PathTracer::fillCameraBasis( cam, out );
	out->focusPoint[0] = -1;
	out->focusPoint[1] = -1;
	out->lense[0] = Cfg::get().value<float>( Cfg::CAM_LENSE_FOCALLENGTH );
	out->lense[1] = Cfg::get().value<float>( Cfg::CAM_LENSE_APERTURE );
}

void pbrh_camera_lookat( const float* eye, const float* center, pbr_camera* out ) {
	Camera cam;
	cam.setEye( eye[0], eye[1], eye[2] );
	cam.setCenter( center[0], center[1], center[2] );
	*out = pbr_camera();
	PathTracer::fillCameraBasis( cam, out );
	out->focusPoint[0] = -1;
	out->focusPoint[1] = -1;
	out->lense[0] = Cfg::get().value<float>( Cfg::CAM_LENSE_FOCALLENGTH );
	out->lense[1] = Cfg::get().value<float>( Cfg::CAM_LENSE_APERTURE );
}

float pbrh_pixel_dimension( uint32_t width, uint32_t height, float fov ) {
	return PathTracer::pixelDimension( width, height, fov );
}

// ---- PathTracer driver -----------------------------------------------------------------

struct HostTracer {
	PathTracer pt;
	Camera cam;
	explicit HostTracer( int device ) : pt( device ) {}
};

void* pbrh_pt_create( int device, uint32_t width, uint32_t height ) {
	try {
		HostTracer* t = new HostTracer( device );
		t->pt.setWidthAndHeight( width, height );
		t->pt.setCamera( &t->cam );
		return t;
	}
	catch( const std::exception& e ) {
		gError = e.what();
		return nullptr;
	}
}

void pbrh_pt_destroy( void* tracer ) {
	delete static_cast<HostTracer*>( tracer );
}

int pbrh_pt_init( void* tracer, void* scene, uint32_t tileWorld, uint32_t tileRank ) {
	HostTracer* t = static_cast<HostTracer*>( tracer );
	HostScene* s = static_cast<HostScene*>( scene );

	try {
		if( s->hasCamera ) {
			t->cam.setEye( s->eye[0], s->eye[1], s->eye[2] );
			t->cam.setCenter( s->center[0], s->center[1], s->center[2] );
		}

		const SceneModel& m = s->loader.getObjParser()->model();
		t->pt.setTiles( tileWorld, tileRank );
		t->pt.initOpenCLBuffers( m.vertices, m.facesV, m.normals, &s->loader, s->bvh.get() );
		return 0;
	}
	catch( const std::exception& e ) {
		gError = e.what();
		return -1;
	}
}

int pbrh_pt_generate_image( void* tracer, float* image, float* debug ) {
	HostTracer* t = static_cast<HostTracer*>( tracer );

	try {
		std::vector<float> dbg;
		const std::vector<float> img = t->pt.generateImage( debug ? &dbg : nullptr );
		std::memcpy( image, img.data(), img.size() * sizeof( float ) );

		if( debug ) {
			std::memcpy( debug, dbg.data(), dbg.size() * sizeof( float ) );
		}

		return 0;
	}
	catch( const std::exception& e ) {
		gError = e.what();
		return -1;
	}
}

int pbrh_pt_generate_images( void* tracer, uint32_t frames, float* image ) {
	HostTracer* t = static_cast<HostTracer*>( tracer );

	try {
		const std::vector<float> img = t->pt.generateImages( frames );
		std::memcpy( image, img.data(), img.size() * sizeof( float ) );
		return 0;
	}
	catch( const std::exception& e ) {
		gError = e.what();
		return -1;
	}
}

void pbrh_pt_set_focus( void* tracer, int x, int y ) {
	static_cast<HostTracer*>( tracer )->pt.setFocus( x, y );
}

void pbrh_pt_reset_sample_count( void* tracer ) {
	static_cast<HostTracer*>( tracer )->pt.resetSampleCount();
}

uint32_t pbrh_pt_sample_count( void* tracer ) {
	return static_cast<HostTracer*>( tracer )->pt.getSampleCount();
}

void* pbrh_pt_context( void* tracer ) {
	return static_cast<HostTracer*>( tracer )->pt.context();
}

void pbrh_pt_camera( void* tracer, pbr_camera* out ) {
	*out = static_cast<HostTracer*>( tracer )->pt.camera();
}


// ---- the CL look-alike, driven the way the reference's PathTracer drives CL ---------------------
// (PathTracer.cpp:136-230 initOpenCLBuffers + initKernelArgs, :59-71 generateImage, :43-52 clPathTracing),
// with the fixed seed sequence seedStep * ( n + 1 ).  Renders `frames` frames of `scene` at the
// configured window size and returns the last accumulated image and debug image.
// refeedEvery > 0: every refeedEvery-th frame feeds the input image twice before executing (a caller that re-uploads the
// same buffer, e.g. after resetting its sample count): the reference uploads twice, the result is the same
int pbrh_cl_adaptor_render_ex( void* scene, uint32_t frames, float seedStep, float* image, float* debug, uint32_t refeedEvery ) {
	HostScene* s = static_cast<HostScene*>( scene );

	try {
		Cfg& cfg = Cfg::get();
		const uint32_t width = cfg.value<uint32_t>( Cfg::WINDOW_WIDTH );
		const uint32_t height = cfg.value<uint32_t>( Cfg::WINDOW_HEIGHT );
		const SceneBuffers& b = s->buffers;
		CL cl( true );
		char msg[128];

		// initOpenCLBuffers_BVH / _Faces / _Materials / _Lights / _Textures
		cl_mem bufBVH = cl.createBuffer( b.bvh, sizeof( pbr_bvh_node ) * b.bvh.size() );
		std::snprintf( msg, sizeof( msg ), "%lu", (unsigned long) b.bvh.size() );
		cl.setReplacement( "#BVH_NUM_NODES#", msg );
		cl_mem bufFacesV = cl.createBuffer( b.facesV, sizeof( pbr_uint4 ) * b.facesV.size() );
		cl_mem bufFacesN = cl.createBuffer( b.facesN, sizeof( pbr_uint4 ) * b.facesN.size() );
		cl_mem bufVertices = cl.createBuffer( b.vertices, sizeof( pbr_float4 ) * b.vertices.size() );
		cl_mem bufNormals = cl.createBuffer( b.normals, sizeof( pbr_float4 ) * b.normals.size() );
		cl_mem bufMaterials = ( b.brdf == 0 )
			? cl.createBuffer( b.materialsSchlick, sizeof( pbr_material_schlick ) * b.materialsSchlick.size() )
			: cl.createBuffer( b.materialsSA, sizeof( pbr_material_sa ) * b.materialsSA.size() );
		std::snprintf( msg, sizeof( msg ), "(float4)( %f, %f, %f, 0.0f )", b.skyLight[0], b.skyLight[1], b.skyLight[2] );
		cl.setReplacement( "#SKY_LIGHT#", msg );
		cl_mem bufLights = cl.createBuffer( b.lights, sizeof( pbr_light ) * b.lights.size() );
		std::snprintf( msg, sizeof( msg ), "%lu", (unsigned long) b.numLights );
		cl.setReplacement( "#NUM_LIGHTS#", msg );

		std::vector<cl_float> textureOut( (size_t) width * height * 4, 0.0f );
		std::vector<cl_float> textureDebug( (size_t) width * height * 4, 0.0f );
		cl_mem bufTextureIn = cl.createImage2DReadOnly( width, height, &textureOut[0] );
		cl_mem bufTextureOut = cl.createImage2DWriteOnly( width, height );
		cl_mem bufTextureDebug = cl.createImage2DWriteOnly( width, height );

		cl.loadProgram( "source/opencl/pathtracing.cl" );
		cl_kernel kernel = cl.createKernel( "pathTracing" );

		// initKernelArgs
		cl_float pxDim = PathTracer::pixelDimension( width, height, cfg.value<float>( Cfg::PERS_FOV ) );
		pbr_camera cam;
		pbrh_scene_camera( scene, &cam );
		cl_uint i = 2;
		cl.setKernelArg( kernel, i++, sizeof( cl_float ), &pxDim );
		cl.setKernelArg( kernel, i++, sizeof( pbr_camera ), &cam );
		cl.setKernelArg( kernel, i++, sizeof( cl_mem ), &bufBVH );
		cl.setKernelArg( kernel, i++, sizeof( cl_mem ), &bufFacesV );
		cl.setKernelArg( kernel, i++, sizeof( cl_mem ), &bufFacesN );
		cl.setKernelArg( kernel, i++, sizeof( cl_mem ), &bufVertices );
		cl.setKernelArg( kernel, i++, sizeof( cl_mem ), &bufNormals );
		cl.setKernelArg( kernel, i++, sizeof( cl_mem ), &bufMaterials );
		cl.setKernelArg( kernel, i++, sizeof( cl_mem ), &bufLights );
		cl.setKernelArg( kernel, i++, sizeof( cl_mem ), &bufTextureIn );
		cl.setKernelArg( kernel, i++, sizeof( cl_mem ), &bufTextureOut );
		cl.setKernelArg( kernel, i++, sizeof( cl_mem ), &bufTextureDebug );

		// generateImage x frames
		for( uint32_t n = 0; n < frames; n++ ) {
			cl.updateImageReadOnly( bufTextureIn, width, height, &textureOut[0] );

			if( refeedEvery > 0 && ( n % refeedEvery ) == refeedEvery - 1 ) {
				cl.updateImageReadOnly( bufTextureIn, width, height, &textureOut[0] );
			}

			cl_float timeSinceStart = seedStep * (float) ( n + 1 );
			cl_float pixelWeight = (float) n / (float) ( n + 1 );
			cl.setKernelArg( kernel, 0, sizeof( cl_float ), &timeSinceStart );
			cl.setKernelArg( kernel, 1, sizeof( cl_float ), &pixelWeight );
			cl.setKernelArg( kernel, 3, sizeof( pbr_camera ), &cam );
			cl.execute( kernel );
			cl.finish();
			cl.readImageOutput( bufTextureOut, width, height, &textureOut[0] );
			cl.readImageOutput( bufTextureDebug, width, height, &textureDebug[0] );
		}

		std::memcpy( image, textureOut.data(), textureOut.size() * sizeof( float ) );
		std::memcpy( debug, textureDebug.data(), textureDebug.size() * sizeof( float ) );
		return 0;
	}
	catch( const std::exception& e ) {
		gError = e.what();
		return -1;
	}
}

int pbrh_cl_adaptor_render( void* scene, uint32_t frames, float seedStep, float* image, float* debug ) {
	return pbrh_cl_adaptor_render_ex( scene, frames, seedStep, image, debug, 0 );
}


// ---- image files (the writer the reference never had — it links DevIL without using it) ----
// PPM (P6) from RGBA8 as pbr_read_display( top_row_first = 1 ) returns it; PFM (PF, little endian, bottom row first —
// the format's own convention and pbr_read_output's) from the linear float image.  Return 0 / -1.
int pbrh_write_ppm( const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height ) {
	FILE* f = ( path != nullptr && rgba8 != nullptr ) ? std::fopen( path, "wb" ) : nullptr;

	if( f == nullptr ) {
		gError = std::string( "cannot write " ) + ( path ? path : "(null)" );
		return -1;
	}

	std::fprintf( f, "P6\n%u %u\n255\n", width, height );
	std::vector<uint8_t> row( (size_t) width * 3 );

	for( uint32_t y = 0; y < height; y++ ) {
		for( uint32_t x = 0; x < width; x++ ) {
			const uint8_t* px = rgba8 + ( (size_t) y * width + x ) * 4;
			row[(size_t) x * 3 + 0] = px[0];
			row[(size_t) x * 3 + 1] = px[1];
			row[(size_t) x * 3 + 2] = px[2];
		}

		std::fwrite( row.data(), 1, row.size(), f );
	}

	return ( std::fclose( f ) == 0 ) ? 0 : -1;
}

int pbrh_write_pfm( const char* path, const float* rgba, uint32_t width, uint32_t height ) {
	FILE* f = ( path != nullptr && rgba != nullptr ) ? std::fopen( path, "wb" ) : nullptr;

	if( f == nullptr ) {
		gError = std::string( "cannot write " ) + ( path ? path : "(null)" );
		return -1;
	}

	std::fprintf( f, "PF\n%u %u\n-1.0\n", width, height );
	std::vector<float> row( (size_t) width * 3 );

	for( uint32_t y = 0; y < height; y++ ) {
		for( uint32_t x = 0; x < width; x++ ) {
			const float* px = rgba + ( (size_t) y * width + x ) * 4;
			row[(size_t) x * 3 + 0] = px[0];
			row[(size_t) x * 3 + 1] = px[1];
			row[(size_t) x * 3 + 2] = px[2];
		}

		std::fwrite( row.data(), sizeof( float ), row.size(), f );
	}

	return ( std::fclose( f ) == 0 ) ? 0 : -1;
}

// PNG (8-bit RGBA, top row first — what pbr_read_display( top_row_first = 1 ) returns) and OpenEXR (binary32 R, G, B + A =
// the accumulated first-hit distance; bottom row first in, as pbr_read_output has it) — the two formats SURVEY.md names
// for this row.  Written here without a library: the PNG's zlib stream uses stored (uncompressed) deflate blocks, the EXR
// is a single-part scanline file with NO_COMPRESSION.  Both are what any reader expects; neither is small.
}  // extern "C"

namespace {

struct Crc32Table {
	uint32_t at[256];

	Crc32Table() {
		for( uint32_t i = 0; i < 256; i++ ) {
			uint32_t c = i;

			for( int k = 0; k < 8; k++ ) {
				c = ( c & 1u ) ? ( 0xEDB88320u ^ ( c >> 1 ) ) : ( c >> 1 );
			}

			at[i] = c;
		}
	}
};

uint32_t crc32Of( const uint8_t* data, size_t n, uint32_t crc ) {
	static const Crc32Table built;        // a function-local static: initialised once, thread-safe (C++11)
	const uint32_t* table = built.at;
	crc = ~crc;

	for( size_t i = 0; i < n; i++ ) {
		crc = table[( crc ^ data[i] ) & 0xFFu] ^ ( crc >> 8 );
	}

	return ~crc;
}

void putBE32( std::vector<uint8_t>& out, uint32_t v ) {
	out.push_back( (uint8_t) ( v >> 24 ) ); out.push_back( (uint8_t) ( v >> 16 ) ); out.push_back( (uint8_t) ( v >> 8 ) ); out.push_back( (uint8_t) v );
}

bool writeChunk( FILE* f, const char type[4], const std::vector<uint8_t>& body ) {
	std::vector<uint8_t> head;
	putBE32( head, (uint32_t) body.size() );
	std::vector<uint8_t> typed( type, type + 4 );
	typed.insert( typed.end(), body.begin(), body.end() );
	std::vector<uint8_t> tail;
	putBE32( tail, crc32Of( typed.data(), typed.size(), 0u ) );
	return std::fwrite( head.data(), 1, 4, f ) == 4 && std::fwrite( typed.data(), 1, typed.size(), f ) == typed.size() && std::fwrite( tail.data(), 1, 4, f ) == 4;
}

template<typename T>
void putLE( std::vector<uint8_t>& out, T v ) {
	uint8_t raw[sizeof( T )];
	std::memcpy( raw, &v, sizeof( T ) );     // the hosts this builds for are little endian, like both file formats' fields here
	out.insert( out.end(), raw, raw + sizeof( T ) );
}

void putAttr( std::vector<uint8_t>& out, const char* name, const char* type, const std::vector<uint8_t>& value ) {
	out.insert( out.end(), name, name + std::strlen( name ) + 1 );
	out.insert( out.end(), type, type + std::strlen( type ) + 1 );
	putLE<int32_t>( out, (int32_t) value.size() );
	out.insert( out.end(), value.begin(), value.end() );
}

}  // namespace

extern "C" {

int pbrh_write_png( const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height ) {
	FILE* f = ( path != nullptr && rgba8 != nullptr && width > 0 && height > 0 ) ? std::fopen( path, "wb" ) : nullptr;

	if( f == nullptr ) {
		gError = std::string( "cannot write " ) + ( path ? path : "(null)" );
		return -1;
	}

	// the filtered image: every scanline = filter type 0 + its RGBA bytes
	std::vector<uint8_t> raw;
	raw.reserve( (size_t) height * ( (size_t) width * 4 + 1 ) );

	for( uint32_t y = 0; y < height; y++ ) {
		raw.push_back( 0 );
		raw.insert( raw.end(), rgba8 + (size_t) y * width * 4, rgba8 + (size_t) ( y + 1 ) * width * 4 );
	}

	// zlib stream: header, stored deflate blocks of <= 65535 bytes, Adler-32 of the raw bytes
	std::vector<uint8_t> z;
	z.push_back( 0x78 ); z.push_back( 0x01 );
	uint32_t a = 1, b = 0;

	for( size_t at = 0; at < raw.size(); ) {
		const size_t n = std::min<size_t>( 65535, raw.size() - at );
		z.push_back( ( at + n == raw.size() ) ? 1 : 0 );
		z.push_back( (uint8_t) ( n & 0xFF ) ); z.push_back( (uint8_t) ( n >> 8 ) );
		z.push_back( (uint8_t) ( ~n & 0xFF ) ); z.push_back( (uint8_t) ( ( ~n >> 8 ) & 0xFF ) );
		z.insert( z.end(), raw.begin() + at, raw.begin() + at + n );

		for( size_t i = at; i < at + n; i++ ) {
			a = ( a + raw[i] ) % 65521u;
			b = ( b + a ) % 65521u;
		}

		at += n;
	}

	putBE32( z, ( b << 16 ) | a );

	const uint8_t signature[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
	std::vector<uint8_t> ihdr;
	putBE32( ihdr, width ); putBE32( ihdr, height );
	ihdr.push_back( 8 ); ihdr.push_back( 6 ); ihdr.push_back( 0 ); ihdr.push_back( 0 ); ihdr.push_back( 0 );   // 8 bits, RGBA, deflate, adaptive, no interlace
	bool ok = std::fwrite( signature, 1, 8, f ) == 8 && writeChunk( f, "IHDR", ihdr ) && writeChunk( f, "IDAT", z ) && writeChunk( f, "IEND", std::vector<uint8_t>() );
	ok = ( std::fclose( f ) == 0 ) && ok;

	if( !ok ) {
		gError = std::string( "short write: " ) + path;
	}

	return ok ? 0 : -1;
}

int pbrh_write_exr( const char* path, const float* rgba, uint32_t width, uint32_t height ) {
	FILE* f = ( path != nullptr && rgba != nullptr && width > 0 && height > 0 ) ? std::fopen( path, "wb" ) : nullptr;

	if( f == nullptr ) {
		gError = std::string( "cannot write " ) + ( path ? path : "(null)" );
		return -1;
	}

	std::vector<uint8_t> head;
	putLE<uint32_t>( head, 20000630u );      // magic
	putLE<uint32_t>( head, 2u );             // version 2, single-part scanline, no flags

	std::vector<uint8_t> channels;           // alphabetical: A, B, G, R; FLOAT (2), linear, no subsampling
	for( const char* name : { "A", "B", "G", "R" } ) {
		channels.insert( channels.end(), name, name + 2 );
		putLE<int32_t>( channels, 2 );
		channels.push_back( 0 ); channels.push_back( 0 ); channels.push_back( 0 ); channels.push_back( 0 );
		putLE<int32_t>( channels, 1 ); putLE<int32_t>( channels, 1 );
	}
	channels.push_back( 0 );
	putAttr( head, "channels", "chlist", channels );
	putAttr( head, "compression", "compression", std::vector<uint8_t>( 1, 0 ) );
	std::vector<uint8_t> window;
	putLE<int32_t>( window, 0 ); putLE<int32_t>( window, 0 ); putLE<int32_t>( window, (int32_t) width - 1 ); putLE<int32_t>( window, (int32_t) height - 1 );
	putAttr( head, "dataWindow", "box2i", window );
	putAttr( head, "displayWindow", "box2i", window );
	putAttr( head, "lineOrder", "lineOrder", std::vector<uint8_t>( 1, 0 ) );      // increasing y: the top row first
	std::vector<uint8_t> one; putLE<float>( one, 1.0f );
	putAttr( head, "pixelAspectRatio", "float", one );
	std::vector<uint8_t> centre; putLE<float>( centre, 0.0f ); putLE<float>( centre, 0.0f );
	putAttr( head, "screenWindowCenter", "v2f", centre );
	putAttr( head, "screenWindowWidth", "float", one );
	head.push_back( 0 );                     // end of the header

	const size_t lineBytes = (size_t) width * 4 * sizeof( float );
	uint64_t offset = head.size() + (uint64_t) height * 8;

	for( uint32_t y = 0; y < height; y++ ) {
		putLE<uint64_t>( head, offset );
		offset += 8 + lineBytes;
	}

	bool ok = std::fwrite( head.data(), 1, head.size(), f ) == head.size();
	std::vector<uint8_t> line;

	for( uint32_t y = 0; ok && y < height; y++ ) {
		// file row y = image row height - 1 - y: the input's row 0 is the bottom of the picture (pbr_read_output)
		const float* row = rgba + (size_t) ( height - 1 - y ) * width * 4;
		line.clear();
		putLE<int32_t>( line, (int32_t) y );
		putLE<int32_t>( line, (int32_t) lineBytes );
		const int order[4] = { 3, 2, 1, 0 };     // A, B, G, R out of R, G, B, A

		for( int c = 0; c < 4; c++ ) {
			for( uint32_t x = 0; x < width; x++ ) {
				putLE<float>( line, row[(size_t) x * 4 + order[c]] );
			}
		}

		ok = std::fwrite( line.data(), 1, line.size(), f ) == line.size();
	}

	ok = ( std::fclose( f ) == 0 ) && ok;

	if( !ok ) {
		gError = std::string( "short write: " ) + path;
	}

	return ok ? 0 : -1;
}

}  // extern "C"
