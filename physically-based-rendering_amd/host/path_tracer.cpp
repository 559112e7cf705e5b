#include "path_tracer.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "Cfg.h"

namespace pbr {

namespace {

// glm::dot / glm::cross / glm::normalize as glm evaluates them for vec3 (products first, summed
// left to right; normalize = v * ( 1 / sqrt( dot ) )), no contraction.
inline float dot3( const float* a, const float* b ) {
	return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

inline void cross3( const float* a, const float* b, float* out ) {
	out[0] = a[1] * b[2] - b[1] * a[2];
	out[1] = a[2] * b[0] - b[2] * a[0];
	out[2] = a[0] * b[1] - b[0] * a[1];
}

inline void normalize3( float* v ) {
	const float inv = 1.0f / std::sqrt( dot3( v, v ) );
	v[0] *= inv;
	v[1] *= inv;
	v[2] *= inv;
}

// The miss link of a container node (PathTracer.cpp:278-307): a left child continues at its
// right sibling; a right child climbs while it is on the right edge of a subtree and continues
// at the right sibling of the first ancestor that is a left child; none => -1 (ends the walk).
// Ids are shifted by the number of skip-ahead deletions before the target.
float missLink( const BVHNode* node ) {
	const BVHNode* parent = node->parent;

	if( parent->leftChild == node ) {
		return (float) ( parent->rightChild->id - parent->rightChild->numSkipsToHere );
	}

	const BVHNode* up = parent;

	while( up->parent != nullptr && up->parent->rightChild == up ) {
		up = up->parent;
	}

	if( up->parent == nullptr ) {
		return -1.0f;
	}

	const BVHNode* target = up->parent->rightChild;

	return (float) ( target->id - target->numSkipsToHere );
}

}  // namespace


// ---------------------------------------------------------------------------
// SceneBuffers
// ---------------------------------------------------------------------------

void SceneBuffers::build( ModelLoader* ml, BVH* accel ) {
	const SceneModel& model = ml->getObjParser()->model();
	brdf = (uint32_t) Cfg::get().value<int>( Cfg::RENDER_BRDF );

	// initOpenCLBuffers_Faces, PathTracer.cpp:357-380: vec3 -> float4, w = 0
	vertices.clear();
	normals.clear();

	for( size_t i = 0; i + 2 < model.vertices.size(); i += 3 ) {
		vertices.push_back( { model.vertices[i], model.vertices[i + 1], model.vertices[i + 2], 0.0f } );
	}
	for( size_t i = 0; i + 2 < model.normals.size(); i += 3 ) {
		normals.push_back( { model.normals[i], model.normals[i + 1], model.normals[i + 2], 0.0f } );
	}

	// initOpenCLBuffers_BVH, PathTracer.cpp:238-347
	bvh.clear();
	facesV.clear();
	facesN.clear();

	const std::vector<BVHNode*>& nodes = accel->nodes();
	bool skipNext = false;

	for( size_t i = 0; i < nodes.size(); i++ ) {
		const BVHNode* node = nodes[i];

		// The left child of a marked node is dropped from the array; a dropped node that is itself
		// marked passes the mark on (PathTracer.cpp:250-256).
		if( skipNext ) {
			skipNext = node->skipNextLeft;
			continue;
		}

		const size_t numFaces = node->faces.size();
		pbr_bvh_node sn;
		sn.bbMin = { node->bbMin[0], node->bbMin[1], node->bbMin[2], ( numFaces > 0 ) ? (float) facesV.size() : -1.0f };
		sn.bbMax = { node->bbMax[0], node->bbMax[1], node->bbMax[2], ( numFaces > 1 ) ? (float) ( facesV.size() + 1 ) : -1.0f };

		if( numFaces == 0 && node->skipNextLeft ) {
			skipNext = true;
		}

		if( node->parent != nullptr && numFaces == 0 ) {
			sn.bbMax.w = missLink( node );
		}

		bvh.push_back( sn );

		// Faces in leaf order, material in .w (PathTracer.cpp:312-330)
		for( size_t j = 0; j < numFaces; j++ ) {
			const Tri& tri = node->faces[j];
			const size_t f = (size_t) tri.face.w * 3;
			const size_t fn = (size_t) tri.normals.w * 3;
			pbr_uint4 fv = { model.facesV.at( f ), model.facesV.at( f + 1 ), model.facesV.at( f + 2 ), (uint32_t) model.facesMtl.at( tri.face.w ) };
			pbr_uint4 fnv = { 0, 0, 0, 0 };

			if( fn + 2 < model.facesVN.size() ) {
				fnv = { model.facesVN[fn], model.facesVN[fn + 1], model.facesVN[fn + 2], 0 };
			}

			facesV.push_back( fv );
			facesN.push_back( fnv );
		}
	}

	// initOpenCLBuffers_MaterialsRGB, PathTracer.cpp:448-519.  SKY_LIGHT travels through
	// snprintf( "%f" ) into the kernel source, i.e. it is rounded to 6 decimals (:470-472).
	materialsSchlick.clear();
	materialsSA.clear();
	skyLight[0] = skyLight[1] = skyLight[2] = 1.0f;
	skyLight[3] = 0.0f;

	for( size_t i = 0; i < model.materials.size(); i++ ) {
		const material_t& m = model.materials[i];
		const pbr_float4 kd = { m.Kd.x, m.Kd.y, m.Kd.z, m.Kd.w };
		const pbr_float4 ks = { m.Ks.x, m.Ks.y, m.Ks.z, m.Ks.w };

		if( brdf == 0 ) {
			materialsSchlick.push_back( { { m.d, m.Ni, m.p, m.rough }, kd, ks } );
		}
		else {
			materialsSA.push_back( { { m.d, m.Ni, m.nu, m.nv, m.Rs, m.Rd, 0.0f, 0.0f }, kd, ks } );
		}

		if( m.mtlName == "sky_light" ) {
			const float rgb[3] = { m.Kd.x, m.Kd.y, m.Kd.z };

			for( int k = 0; k < 3; k++ ) {
				char text[64];
				snprintf( text, sizeof( text ), "%f", rgb[k] );
				skyLight[k] = (float) strtod( text, nullptr );
			}
		}
	}

	// initOpenCLBuffers_Lights, PathTracer.cpp:387-428
	lights.clear();

	for( size_t i = 0; i < model.lights.size(); i++ ) {
		const light_t& l = model.lights[i];
		pbr_light out;
		out.pos = { l.pos.x, l.pos.y, l.pos.z, l.pos.w };
		out.rgb = { l.rgb.x, l.rgb.y, l.rgb.z, l.rgb.w };
		out.data = { (float) l.type, ( l.type == 2 ) ? l.radius : 0.0f, 0.0f, 0.0f };
		lights.push_back( out );
	}

	numLights = (uint32_t) lights.size();

	if( lights.empty() ) {
		pbr_light dummy = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
		lights.push_back( dummy );
	}
}


pbr_scene_desc SceneBuffers::desc() const {
	pbr_scene_desc d;
	d.bvh = bvh.data();
	d.num_nodes = (uint32_t) bvh.size();
	d.facesV = facesV.data();
	d.facesN = facesN.data();
	d.num_faces = (uint32_t) facesV.size();
	d.vertices = vertices.data();
	d.num_vertices = (uint32_t) vertices.size();
	d.normals = normals.data();
	d.num_normals = (uint32_t) normals.size();
	d.brdf = brdf;
	d.materials = ( brdf == 0 ) ? (const void*) materialsSchlick.data() : (const void*) materialsSA.data();
	d.num_materials = (uint32_t) ( ( brdf == 0 ) ? materialsSchlick.size() : materialsSA.size() );
	d.lights = lights.data();
	d.num_lights = numLights;
	return d;
}


// ---------------------------------------------------------------------------
// Camera
// ---------------------------------------------------------------------------

Camera::Camera() {
	this->cameraReset();
}

// Camera::cameraReset, Camera.cpp:80-95
void Camera::cameraReset() {
	mEye[0] = Cfg::get().value<float>( Cfg::CAM_EYE_X );
	mEye[1] = Cfg::get().value<float>( Cfg::CAM_EYE_Y );
	mEye[2] = Cfg::get().value<float>( Cfg::CAM_EYE_Z );
	mUp[0] = 0.0f;
	mUp[1] = 1.0f;
	mUp[2] = 0.0f;
	this->setCenter(
		Cfg::get().value<float>( Cfg::CAM_CENTER_X ),
		Cfg::get().value<float>( Cfg::CAM_CENTER_Y ),
		Cfg::get().value<float>( Cfg::CAM_CENTER_Z )
	);
}

void Camera::setCenter( float x, float y, float z ) {
	mCenter[0] = x;
	mCenter[1] = y;
	mCenter[2] = z;
	normalize3( mCenter );
}

// Camera::getAdjustedCenter_glmVec3, Camera.cpp:102-108
void Camera::getAdjustedCenter( float out[3] ) const {
	out[0] = mEye[0] + mCenter[0];
	out[1] = mEye[1] - mCenter[1];
	out[2] = mEye[2] - mCenter[2];
}


// ---------------------------------------------------------------------------
// PathTracer
// ---------------------------------------------------------------------------

// PathTracer::PathTracer, PathTracer.cpp:11-28
PathTracer::PathTracer( int device ) {
	mDevice = device;
	mWidth = Cfg::get().value<uint32_t>( Cfg::WINDOW_WIDTH );
	mHeight = Cfg::get().value<uint32_t>( Cfg::WINDOW_HEIGHT );
	mFOV = Cfg::get().value<float>( Cfg::PERS_FOV );
	mSampleCount = 0;
	mSeedStep = 0.0333f;
	mTileWorld = 1;
	mTileRank = 0;
	mCamera = nullptr;
	mCtx = nullptr;

	mStructCam = pbr_camera();
	mStructCam.focusPoint[0] = -1;
	mStructCam.focusPoint[1] = -1;
	mStructCam.lense[0] = Cfg::get().value<float>( Cfg::CAM_LENSE_FOCALLENGTH );
	mStructCam.lense[1] = Cfg::get().value<float>( Cfg::CAM_LENSE_APERTURE );
}

PathTracer::~PathTracer() {
	if( mCtx != nullptr ) {
		pbr_destroy( mCtx );
	}
}

void PathTracer::check( int status, const char* what ) {
	if( status != PBR_OK ) {
		const char* msg = ( mCtx != nullptr ) ? pbr_last_error( mCtx ) : "no context";
		throw std::runtime_error( std::string( "[PathTracer] " ) + what + ": " + msg );
	}
}

// initKernelArgs, PathTracer.cpp:89-91; MathHelp::degToRad, MathHelp.cpp:9-11 (double constant)
float PathTracer::pixelDimension( uint32_t width, uint32_t height, float fovDegrees ) {
	const float aspect = (float) width / (float) height;
	const float rad = (float) ( fovDegrees * 3.14159265359 / 180.0f );
	const float f = aspect * 2.0f * (float) std::tan( (double) ( rad / 2.0f ) );
	return f / (float) width;
}

// updateEyeBuffer, PathTracer.cpp:628-652
void PathTracer::fillCameraBasis( const Camera& cam, pbr_camera* out ) {
	float c[3];
	cam.getAdjustedCenter( c );
	const float* eye = cam.getEye();
	const float* up = cam.getUp();

	float w[3] = { c[0] - eye[0], c[1] - eye[1], c[2] - eye[2] };
	normalize3( w );
	float u[3];
	cross3( w, up, u );
	normalize3( u );
	float v[3];
	cross3( u, w, v );
	normalize3( v );

	out->eye = { eye[0], eye[1], eye[2], 0.0f };
	out->w = { w[0], w[1], w[2], 0.0f };
	out->u = { u[0], u[1], u[2], 0.0f };
	out->v = { v[0], v[1], v[2], 0.0f };
}

void PathTracer::updateEyeBuffer() {
	if( mCamera != nullptr ) {
		fillCameraBasis( *mCamera, &mStructCam );
	}
}

pbr_config PathTracer::makeConfig( const SceneBuffers& buffers, uint32_t width, uint32_t height ) {
	pbr_config cfg = pbr_config();
	cfg.width = width;
	cfg.height = height;
	cfg.brdf = buffers.brdf;
	cfg.shadow_rays = Cfg::get().value<uint32_t>( Cfg::RENDER_SHADOWRAYS );
	cfg.max_depth = Cfg::get().value<uint32_t>( Cfg::RENDER_MAXDEPTH );
	cfg.max_added_depth = Cfg::get().value<uint32_t>( Cfg::RENDER_MAXADDEDDEPTH );
	cfg.samples = Cfg::get().value<uint32_t>( Cfg::RENDER_SAMPLES );
	cfg.anti_aliasing = Cfg::get().value<float>( Cfg::RENDER_ANTIALIAS );
	cfg.phong_tessellation = Cfg::get().value<float>( Cfg::RENDER_PHONGTESS );

	for( int k = 0; k < 4; k++ ) {
		cfg.sky_light[k] = buffers.skyLight[k];
	}

	cfg.tile_world = 1;
	cfg.tile_rank = 0;
	cfg.traversal = Cfg::get().value<uint32_t>( Cfg::HIP_TRAVERSAL );   // 0 unless the caller's configuration asks for a mode
	cfg.arith = Cfg::get().value<uint32_t>( Cfg::HIP_ARITH );
	return cfg;
}

void PathTracer::initOpenCLBuffers(
	std::vector<float> vertices, std::vector<uint32_t> faces, std::vector<float> normals,
	ModelLoader* ml, AccelStructure* accel
) {
	// The reference receives vertices / faces / normals separately although they are the loader's
	// own arrays (GLWidget.cpp:379); the loader is the single source here.
	(void) vertices;
	(void) faces;
	(void) normals;

	if( mCtx != nullptr ) {
		pbr_destroy( mCtx );
		mCtx = nullptr;
	}

	mBuffers.build( ml, static_cast<BVH*>( accel ) );

	pbr_ctx* ctx = nullptr;

	if( pbr_create( mDevice, &ctx ) != PBR_OK ) {
		std::string msg = ( ctx != nullptr ) ? pbr_last_error( ctx ) : "pbr_create failed";

		if( ctx != nullptr ) {
			pbr_destroy( ctx );
		}

		throw std::runtime_error( "[PathTracer] " + msg );
	}

	mCtx = ctx;

	const pbr_scene_desc desc = mBuffers.desc();
	this->check( pbr_upload_scene( mCtx, &desc ), "pbr_upload_scene" );

	pbr_config cfg = makeConfig( mBuffers, mWidth, mHeight );
	cfg.tile_world = mTileWorld;
	cfg.tile_rank = mTileRank;
	this->check( pbr_configure( mCtx, &cfg ), "pbr_configure" );

	mSampleCount = 0;
}

std::vector<float> PathTracer::generateImage( std::vector<float>* textureDebug ) {
	if( mCtx == nullptr ) {
		throw std::runtime_error( "[PathTracer] generateImage before initOpenCLBuffers" );
	}

	this->updateEyeBuffer();

	const float seed = mSeedStep * (float) ( mSampleCount + 1 );
	const float pixelWeight = mSampleCount / (float) ( mSampleCount + 1 );
	const float pxDim = pixelDimension( mWidth, mHeight, mFOV );

	this->check( pbr_render_frame( mCtx, seed, pixelWeight, pxDim, &mStructCam ), "pbr_render_frame" );

	std::vector<float> image( (size_t) mWidth * mHeight * 4 );
	this->check( pbr_read_output( mCtx, image.data() ), "pbr_read_output" );

	if( textureDebug != nullptr ) {
		textureDebug->resize( image.size() );
		this->check( pbr_read_debug( mCtx, textureDebug->data() ), "pbr_read_debug" );
	}

	this->check( pbr_accumulate( mCtx ), "pbr_accumulate" );
	mSampleCount++;

	return image;
}

std::vector<float> PathTracer::generateImages( uint32_t frames ) {
	if( mCtx == nullptr ) {
		throw std::runtime_error( "[PathTracer] generateImages before initOpenCLBuffers" );
	}

	this->updateEyeBuffer();

	std::vector<float> seeds( frames );

	for( uint32_t k = 0; k < frames; k++ ) {
		seeds[k] = mSeedStep * (float) ( mSampleCount + k + 1 );
	}

	const float pxDim = pixelDimension( mWidth, mHeight, mFOV );
	this->check( pbr_render( mCtx, mSampleCount, frames, seeds.data(), pxDim, &mStructCam ), "pbr_render" );
	mSampleCount += frames;

	std::vector<float> image( (size_t) mWidth * mHeight * 4 );
	this->check( pbr_read_output( mCtx, image.data() ), "pbr_read_output" );

	return image;
}

// PathTracer::resetSampleCount, PathTracer.cpp:576-578.  The image is NOT cleared: the next
// frame's pixelWeight is 0, which overwrites the colour, and depth of field still needs the
// previous frame's hit distances in .w (pathtracing.cl:58-65).
void PathTracer::resetSampleCount() {
	mSampleCount = 0;
}

// PathTracer::setFocus, PathTracer.cpp:596-602
void PathTracer::setFocus( int x, int y ) {
	mStructCam.focusPoint[0] = x;
	mStructCam.focusPoint[1] = y;
	mSampleCount = 0;
}

double PathTracer::getKernelTime() const {
	return ( mCtx != nullptr ) ? pbr_last_kernel_ms( mCtx ) : 0.0;
}

}  // namespace pbr
