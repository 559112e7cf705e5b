// extern "C" surface of libpbrhost.so for the Python test / bench harness (ctypes).
// It exposes the host half of the path — config, scene load / generation, BVH build,
// buffer packing, camera, PathTracer driver — with plain pointers, so the harness can hand the
// SAME flat arrays to the HIP core (libpbrhip.so) and to the CPU oracle.
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

}  // extern "C"
