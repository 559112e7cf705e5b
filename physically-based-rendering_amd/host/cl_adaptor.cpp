#include "cl_adaptor.h"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "Cfg.h"

using std::string;

namespace {

// kernel-argument slots of `pathTracing` (PathTracer.cpp:43-48,97-124)
enum {
	ARG_SEED = 0, ARG_WEIGHT = 1, ARG_PXDIM = 2, ARG_CAMERA = 3, ARG_BVH = 4, ARG_FACES_V = 5, ARG_FACES_N = 6,
	ARG_VERTICES = 7, ARG_NORMALS = 8, ARG_MATERIALS = 9, ARG_LIGHTS = 10, ARG_IMAGE_IN = 11, ARG_IMAGE_OUT = 12, ARG_IMAGE_DEBUG = 13
};

}  // namespace


CL::CL( const bool silent )
	: mSilent( silent ), mCtx( nullptr ), mKernelTag( 0 ), mProgramLoaded( false ), mSceneDirty( true ), mInputDirty( false ),
	  mLastReadTarget( nullptr ) {
	// the work size is fixed at construction from the configured window size (CL.cpp:18-19)
	mWorkWidth = pbr::Cfg::get().value<uint32_t>( pbr::Cfg::WINDOW_WIDTH );
	mWorkHeight = pbr::Cfg::get().value<uint32_t>( pbr::Cfg::WINDOW_HEIGHT );

	if( pbr_create( 0, &mCtx ) != PBR_OK ) {
		const string msg = string( "[CL] pbr_create: " ) + pbr_last_error( mCtx );
		pbr_destroy( mCtx );
		mCtx = nullptr;
		throw std::runtime_error( msg );
	}

	if( !mSilent ) {
		std::fprintf( stderr, "[CL] HIP core behind the CL interface, work size %u x %u\n", mWorkWidth, mWorkHeight );
	}
}


CL::~CL() {
	pbr_destroy( mCtx );
}


void CL::check( int status, const char* what ) {
	if( status != PBR_OK ) {
		throw std::runtime_error( string( "[CL] " ) + what + ": " + pbr_last_error( mCtx ) );
	}
}


cl_mem CL::adoptBytes( const void* data, size_t size ) {
	std::unique_ptr<Blob> blob( new Blob() );
	blob->bytes.resize( size );

	if( data != nullptr && size > 0 ) {
		std::memcpy( blob->bytes.data(), data, size );
	}

	mBlobs.push_back( std::move( blob ) );
	mSceneDirty = true;
	return (cl_mem) mBlobs.back().get();
}


CL::Blob* CL::blobOf( cl_mem handle ) {
	for( auto& blob : mBlobs ) {
		if( (cl_mem) blob.get() == handle ) {
			return blob.get();
		}
	}

	throw std::runtime_error( "[CL] unknown cl_mem handle" );
}


cl_mem CL::createEmptyBuffer( size_t size, cl_mem_flags ) {
	return this->adoptBytes( nullptr, size );
}


cl_mem CL::createImage2DReadOnly( size_t width, size_t height, cl_float* data ) {
	const cl_mem handle = this->adoptBytes( data, width * height * 4 * sizeof( cl_float ) );
	Blob* blob = this->blobOf( handle );
	blob->isImage = true;
	blob->width = width;
	blob->height = height;
	mInputDirty = true;
	return handle;
}


cl_mem CL::createImage2DWriteOnly( size_t width, size_t height ) {
	const cl_mem handle = this->adoptBytes( nullptr, 0 );
	Blob* blob = this->blobOf( handle );
	blob->isImage = true;
	blob->writeOnly = true;
	blob->width = width;
	blob->height = height;
	return handle;
}


cl_kernel CL::createKernel( const char* functionName ) {
	if( functionName == nullptr || string( functionName ) != "pathTracing" ) {
		throw std::runtime_error( string( "[CL] createKernel: only `pathTracing` exists, not `" ) + ( functionName ? functionName : "" ) + "`" );
	}
	if( !mProgramLoaded ) {
		throw std::runtime_error( "[CL] createKernel before loadProgram" );
	}

	return (cl_kernel) &mKernelTag;
}


void CL::loadProgram( string ) {
	// nothing to compile: the kernel variants are built ahead of time; the values CL::setValues would
	// paste into the source (CL.cpp:626-705) are read from Cfg when the scene is uploaded
	mProgramLoaded = true;
	mSceneDirty = true;
}


void CL::setReplacement( string before, string after ) {
	mReplacements[before] = after;
	mSceneDirty = true;
}


void CL::setKernelArg( cl_kernel kernel, cl_uint index, size_t size, void* data ) {
	if( kernel != (cl_kernel) &mKernelTag || data == nullptr ) {
		throw std::runtime_error( "[CL] setKernelArg: unknown kernel or null data" );
	}

	std::vector<unsigned char>& slot = mArgs[index];
	const bool changed = ( slot.size() != size ) || std::memcmp( slot.data(), data, size ) != 0;
	slot.assign( (const unsigned char*) data, (const unsigned char*) data + size );

	if( changed && index >= ARG_BVH && index <= ARG_LIGHTS ) {
		mSceneDirty = true;
	}
}


const CL::Blob* CL::argBlob( cl_uint index ) {
	const auto it = mArgs.find( index );

	if( it == mArgs.end() || it->second.size() != sizeof( cl_mem ) ) {
		char msg[96];
		std::snprintf( msg, sizeof( msg ), "[CL] kernel argument %u (a cl_mem) was never set", index );
		throw std::runtime_error( msg );
	}

	cl_mem handle;
	std::memcpy( &handle, it->second.data(), sizeof( handle ) );
	return this->blobOf( handle );
}


// createBuffer x 7 + the substitutions -> pbr_upload_scene + pbr_configure
void CL::uploadScene() {
	pbr::Cfg& cfg = pbr::Cfg::get();
	const uint32_t brdf = cfg.value<uint32_t>( pbr::Cfg::RENDER_BRDF );
	const Blob* bvh = this->argBlob( ARG_BVH );
	const Blob* facesV = this->argBlob( ARG_FACES_V );
	const Blob* facesN = this->argBlob( ARG_FACES_N );
	const Blob* vertices = this->argBlob( ARG_VERTICES );
	const Blob* normals = this->argBlob( ARG_NORMALS );
	const Blob* materials = this->argBlob( ARG_MATERIALS );
	const Blob* lights = this->argBlob( ARG_LIGHTS );

	pbr_scene_desc scene;
	std::memset( &scene, 0, sizeof( scene ) );
	scene.bvh = (const pbr_bvh_node*) bvh->bytes.data();
	scene.num_nodes = (uint32_t) ( bvh->bytes.size() / sizeof( pbr_bvh_node ) );
	scene.facesV = (const pbr_uint4*) facesV->bytes.data();
	scene.facesN = (const pbr_uint4*) facesN->bytes.data();
	scene.num_faces = (uint32_t) ( facesV->bytes.size() / sizeof( pbr_uint4 ) );
	scene.vertices = (const pbr_float4*) vertices->bytes.data();
	scene.num_vertices = (uint32_t) ( vertices->bytes.size() / sizeof( pbr_float4 ) );
	scene.normals = (const pbr_float4*) normals->bytes.data();
	scene.num_normals = (uint32_t) ( normals->bytes.size() / sizeof( pbr_float4 ) );
	scene.brdf = brdf;
	scene.materials = materials->bytes.data();
	scene.num_materials = (uint32_t) ( materials->bytes.size() / ( brdf == 0 ? sizeof( pbr_material_schlick ) : sizeof( pbr_material_sa ) ) );
	scene.lights = (const pbr_light*) lights->bytes.data();

	// #NUM_LIGHTS# / #BVH_NUM_NODES# arrive as decimal text (PathTracer.cpp:209-210,337-338)
	const auto numLights = mReplacements.find( "#NUM_LIGHTS#" );
	scene.num_lights = ( numLights != mReplacements.end() ) ? (uint32_t) std::strtoul( numLights->second.c_str(), nullptr, 10 ) : 0u;
	const auto numNodes = mReplacements.find( "#BVH_NUM_NODES#" );

	if( numNodes != mReplacements.end() ) {
		const uint32_t n = (uint32_t) std::strtoul( numNodes->second.c_str(), nullptr, 10 );
		scene.num_nodes = ( n < scene.num_nodes ) ? n : scene.num_nodes;
	}

	this->check( pbr_upload_scene( mCtx, &scene ), "pbr_upload_scene" );

	pbr_config config;
	std::memset( &config, 0, sizeof( config ) );
	config.width = mWorkWidth;
	config.height = mWorkHeight;
	config.brdf = brdf;
	config.shadow_rays = cfg.value<uint32_t>( pbr::Cfg::RENDER_SHADOWRAYS );
	config.max_depth = cfg.value<uint32_t>( pbr::Cfg::RENDER_MAXDEPTH );
	config.max_added_depth = cfg.value<uint32_t>( pbr::Cfg::RENDER_MAXADDEDDEPTH );
	config.samples = cfg.value<uint32_t>( pbr::Cfg::RENDER_SAMPLES );
	config.anti_aliasing = cfg.value<float>( pbr::Cfg::RENDER_ANTIALIAS );
	config.phong_tessellation = cfg.value<float>( pbr::Cfg::RENDER_PHONGTESS );
	config.sky_light[0] = config.sky_light[1] = config.sky_light[2] = 1.0f;
	config.sky_light[3] = 0.0f;
	config.tile_world = 1;
	config.tile_rank = 0;
	config.traversal = cfg.value<uint32_t>( pbr::Cfg::HIP_TRAVERSAL );
	config.arith = cfg.value<uint32_t>( pbr::Cfg::HIP_ARITH );

	// "(float4)( r, g, b, 0.0f )", the numbers printed with %f (PathTracer.cpp:466-472,495-501,515): what the
	// OpenCL compiler would have read is what is used here
	const auto sky = mReplacements.find( "#SKY_LIGHT#" );

	if( sky != mReplacements.end() ) {
		const string& text = sky->second;
		size_t pos = text.find( ')' );
		pos = ( pos == string::npos ) ? 0 : pos + 1;
		int got = 0;

		while( got < 4 && pos < text.size() ) {
			const char c = text[pos];

			if( ( c >= '0' && c <= '9' ) || c == '-' || c == '+' || c == '.' ) {
				char* end = nullptr;
				config.sky_light[got++] = std::strtof( text.c_str() + pos, &end );
				pos = (size_t) ( end - text.c_str() );
			}
			else {
				pos++;
			}
		}
	}

	this->check( pbr_configure( mCtx, &config ), "pbr_configure" );
	mSceneDirty = false;
	mInputDirty = true;   // pbr_configure cleared the images
}


void CL::execute( cl_kernel kernel ) {
	if( kernel != (cl_kernel) &mKernelTag ) {
		throw std::runtime_error( "[CL] execute: unknown kernel" );
	}

	if( mSceneDirty ) {
		this->uploadScene();
	}

	if( mInputDirty ) {
		const Blob* in = this->argBlob( ARG_IMAGE_IN );

		if( in->bytes.size() == (size_t) mWorkWidth * mWorkHeight * 4 * sizeof( float ) ) {
			this->check( pbr_write_input( mCtx, (const float*) in->bytes.data() ), "pbr_write_input" );
		}

		mInputDirty = false;
	}

	float seed = 0.0f, weight = 0.0f, pxDim = 0.0f;
	pbr_camera camera;
	const auto need = [&]( cl_uint index, void* dst, size_t size ) {
		const auto it = mArgs.find( index );

		if( it == mArgs.end() || it->second.size() != size ) {
			char msg[96];
			std::snprintf( msg, sizeof( msg ), "[CL] kernel argument %u missing or of the wrong size", index );
			throw std::runtime_error( msg );
		}

		std::memcpy( dst, it->second.data(), size );
	};
	need( ARG_SEED, &seed, sizeof( seed ) );
	need( ARG_WEIGHT, &weight, sizeof( weight ) );
	need( ARG_PXDIM, &pxDim, sizeof( pxDim ) );
	need( ARG_CAMERA, &camera, sizeof( camera ) );

	this->check( pbr_render_frame( mCtx, seed, weight, pxDim, &camera ), "pbr_render_frame" );
	mOutputIsFresh = false;   // a new imageOut: nothing the caller holds matches it until the next readImageOutput
}


void CL::finish() {
	// pbr_render_frame returns after the launch has completed
}


void CL::freeBuffers() {
	mBlobs.clear();
	mArgs.clear();
	mSceneDirty = true;
}


std::map<cl_kernel, string> CL::getKernelNames() {
	std::map<cl_kernel, string> names;
	names[(cl_kernel) &mKernelTag] = "pathTracing";
	return names;
}


std::map<cl_kernel, double> CL::getKernelTimes() {
	std::map<cl_kernel, double> times;
	times[(cl_kernel) &mKernelTag] = pbr_last_kernel_ms( mCtx );
	return times;
}


void CL::readImageOutput( cl_mem image, size_t width, size_t height, cl_float* outputTarget ) {
	if( width != mWorkWidth || height != mWorkHeight || outputTarget == nullptr ) {
		throw std::runtime_error( "[CL] readImageOutput: size differs from the work size" );
	}

	const Blob* blob = this->blobOf( image );
	const bool isDebug = ( mArgs.count( ARG_IMAGE_DEBUG ) != 0 ) && ( this->argBlob( ARG_IMAGE_DEBUG ) == blob );

	if( isDebug ) {
		this->check( pbr_read_debug( mCtx, outputTarget ), "pbr_read_debug" );
		return;
	}

	this->check( pbr_read_output( mCtx, outputTarget ), "pbr_read_output" );
	mLastReadTarget = outputTarget;
	mLastRead.assign( outputTarget, outputTarget + width * height * 4 );
	mOutputIsFresh = true;   // imageOut on the device is what the caller now holds: feeding it back is one swap
}


cl_mem CL::updateBuffer( cl_mem buffer, size_t size, void* data ) {
	Blob* blob = this->blobOf( buffer );
	blob->bytes.assign( (const unsigned char*) data, (const unsigned char*) data + size );
	mSceneDirty = true;
	return buffer;
}


cl_mem CL::updateImageReadOnly( cl_mem image, size_t width, size_t height, cl_float* data ) {
	Blob* blob = this->blobOf( image );
	const size_t count = width * height * 4;

	// The reference feeds last frame's output back as this frame's input (PathTracer.cpp:61-67).
	// When that is exactly what `data` holds, the image is already on the device.
	// Once: the swap turns imageOut into imageIn, so a second feed of the same buffer without a frame in between
	// (reset the sample count, then re-feed) must not swap back — it uploads like the reference does.
	if( !mSceneDirty && mOutputIsFresh && data == mLastReadTarget && mLastRead.size() == count &&
	    std::memcmp( data, mLastRead.data(), count * sizeof( float ) ) == 0 ) {
		this->check( pbr_accumulate( mCtx ), "pbr_accumulate" );
		mOutputIsFresh = false;
		mInputDirty = false;
		return image;
	}

	mOutputIsFresh = false;

	blob->bytes.assign( (const unsigned char*) data, (const unsigned char*) data + count * sizeof( float ) );
	mInputDirty = true;
	return image;
}
