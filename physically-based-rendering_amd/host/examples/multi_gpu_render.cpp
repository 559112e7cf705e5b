// INTEGRATION.md section 5 as a program: what a viewer that holds ONE PathTracer does to render on every GPU of the node.
// The reference builds its scene arrays once (ModelLoader / BVH / PathTracer::initOpenCLBuffers_*, PathTracer.cpp:238-519)
// and hands them to one CL*; here the same arrays go to one pbr_multi, which owns a context and a host thread per device.
//
//   g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I physically-based-rendering_amd/host -I /opt/rocm/include \
//       physically-based-rendering_amd/host/examples/multi_gpu_render.cpp -o multi_gpu_render \
//       -L physically-based-rendering_amd/host -lpbrmulti -lpbrhost -L physically-based-rendering_amd/csrc -lpbrhip \
//       -L /opt/rocm/lib -lrccl -lamdhip64 -pthread
//   ./multi_gpu_render [scene [frames [width height]]]          (scripts/check_integration_build.sh compiles and links it)
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "Cfg.h"
#include "bvh_builder.h"
#include "model_io.h"
#include "path_tracer.h"
#include "pbr_multi.h"
#include "scene_gen.h"

using namespace pbr;

static void check( pbr_multi* m, int status, const char* what ) {
	if( status != PBR_OK ) {
		std::fprintf( stderr, "%s: %s\n", what, pbr_multi_last_error( m ) );
		std::exit( EXIT_FAILURE );
	}
}

int main( int argc, char** argv ) {
	const std::string kind = ( argc > 1 ) ? argv[1] : "cornell";
	const uint32_t frames = ( argc > 2 ) ? (uint32_t) std::atoi( argv[2] ) : 64u;
	const uint32_t width = ( argc > 4 ) ? (uint32_t) std::atoi( argv[3] ) : 1920u;
	const uint32_t height = ( argc > 4 ) ? (uint32_t) std::atoi( argv[4] ) : 1080u;

	// the host side "as before": model, BVH, the flat arrays of initOpenCLBuffers_*
	GeneratedScene g = generateScene( kind, 1, 0 );
	ModelLoader loader;
	loader.getObjParser()->adopt( g.model );
	const SceneModel& model = loader.getObjParser()->model();
	BVH bvh( model.objects, model.vertices, model.normals );
	SceneBuffers buffers;
	buffers.build( &loader, &bvh );
	const pbr_scene_desc scene = buffers.desc();
	const pbr_config cfg = PathTracer::makeConfig( buffers, width, height );     // tile_world / tile_rank are filled per context
	Camera camera;
	camera.setEye( g.eye[0], g.eye[1], g.eye[2] );
	camera.setCenter( g.center[0], g.center[1], g.center[2] );
	pbr_camera cam = pbr_camera();
	PathTracer::fillCameraBasis( camera, &cam );
	cam.focusPoint[0] = cam.focusPoint[1] = -1;
	cam.lense[0] = Cfg::get().value<float>( Cfg::CAM_LENSE_FOCALLENGTH );
	cam.lense[1] = Cfg::get().value<float>( Cfg::CAM_LENSE_APERTURE );
	const float pxDim = PathTracer::pixelDimension( width, height, 45.0f );

	// every GPU of the node
	int count = 0;

	if( hipGetDeviceCount( &count ) != hipSuccess || count < 1 ) {
		std::fprintf( stderr, "no HIP device\n" );
		return EXIT_FAILURE;
	}

	std::vector<int> devices( (size_t) count );

	for( int d = 0; d < count; d++ ) {
		devices[(size_t) d] = d;
	}

	pbr_multi* m = nullptr;
	check( nullptr, pbr_multi_create( devices.data(), count, PBR_MULTI_RCCL, &m ), "pbr_multi_create" );
	check( m, pbr_multi_upload_scene( m, &scene ), "pbr_multi_upload_scene" );
	check( m, pbr_multi_configure( m, &cfg ), "pbr_multi_configure" );

	int plan = -1;
	check( m, pbr_multi_tune( m, frames, pxDim, &cam, &plan, nullptr ), "pbr_multi_tune" );

	std::vector<float> seeds( frames );

	for( uint32_t k = 0; k < frames; k++ ) {
		seeds[k] = 0.0333f * (float) ( k + 1u );
	}

	check( m, pbr_multi_render( m, 0, frames, seeds.data(), pxDim, &cam, 1 ), "pbr_multi_render" );
	std::vector<double> renderMs( (size_t) count ), gatherMs( (size_t) count );
	pbr_multi_timings( m, renderMs.data(), gatherMs.data() );
	std::vector<float> image( (size_t) width * height * 4 );
	check( m, pbr_multi_read_full( m, 0, image.data() ), "pbr_multi_read_full" );

	double sum[3] = { 0.0, 0.0, 0.0 };

	for( size_t p = 0; p < (size_t) width * height; p++ ) {
		for( int c = 0; c < 3; c++ ) {
			sum[c] += image[p * 4 + (size_t) c];
		}
	}

	std::printf( "%s %ux%u, %u frames on %d GPU(s), schedule %d: mean colour %.5f %.5f %.5f\n", kind.c_str(), width, height, frames, count, plan,
	             sum[0] / ( width * height ), sum[1] / ( width * height ), sum[2] / ( width * height ) );

	for( int r = 0; r < count; r++ ) {
		std::printf( "  rank %d: render %.3f ms, exchange %.3f ms\n", r, renderMs[(size_t) r], gatherMs[(size_t) r] );
	}

	pbr_multi_destroy( m );
	return EXIT_SUCCESS;
}
