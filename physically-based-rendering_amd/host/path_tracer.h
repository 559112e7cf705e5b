// Render driver: the host half of the hot path.  Mirrors the reference's PathTracer
// (source/PathTracer.h:80-146) and the part of Camera it reads (source/Camera.h,
// Camera.cpp:80-107) with the same method names, but talks to the HIP core through the C ABI
// (include/pbr_hip.h) instead of `CL`, and keeps the accumulated image on the device.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "bvh_builder.h"
#include "model_io.h"
#include "pbr_hip.h"

namespace pbr {

// The flat arrays PathTracer::initOpenCLBuffers_* upload (PathTracer.cpp:238-533), in the
// reference's wire formats.  Pure host data: built and testable without a device.
struct SceneBuffers {
	std::vector<pbr_bvh_node> bvh;
	std::vector<pbr_uint4> facesV;
	std::vector<pbr_uint4> facesN;
	std::vector<pbr_float4> vertices;
	std::vector<pbr_float4> normals;
	std::vector<pbr_material_schlick> materialsSchlick;  // filled when brdf == 0
	std::vector<pbr_material_sa> materialsSA;            // filled when brdf == 1
	std::vector<pbr_light> lights;                       // >= 1 entry (dummy when the scene has none)
	uint32_t numLights = 0;                              // #NUM_LIGHTS#
	uint32_t brdf = 1;
	float skyLight[4] = { 1.0f, 1.0f, 1.0f, 0.0f };      // #SKY_LIGHT#

	// initOpenCLBuffers_Faces / _BVH / _Materials / _Lights in the reference's order
	// (PathTracer.cpp:164-210).  Reads render.brdf from Cfg.
	void build( ModelLoader* ml, BVH* bvh );
	pbr_scene_desc desc() const;
};


// camera_t subset + Camera::cameraReset / getAdjustedCenter_glmVec3 (Camera.cpp:80-107).
class Camera {

	public:
		Camera();
		void cameraReset();
		void setEye( float x, float y, float z ) { mEye[0] = x; mEye[1] = y; mEye[2] = z; }
		// `center` is a view direction in the reference's convention: target = eye + (cx, -cy, -cz).
		void setCenter( float x, float y, float z );
		void getAdjustedCenter( float out[3] ) const;
		const float* getEye() const { return mEye; }
		const float* getUp() const { return mUp; }

	private:
		float mEye[3], mCenter[3], mUp[3];

};


class PathTracer {

	public:
		// `device`: HIP device ordinal for the context created in initOpenCLBuffers.
		explicit PathTracer( int device = 0 );
		~PathTracer();

		// PathTracer::initOpenCLBuffers (PathTracer.cpp:136-230): new context, upload, configure.
		// Throws std::runtime_error with the C ABI's message on failure (the reference exit()s).
		void initOpenCLBuffers(
			std::vector<float> vertices, std::vector<uint32_t> faces, std::vector<float> normals,
			ModelLoader* ml, AccelStructure* bvh
		);
		// PathTracer::generateImage (PathTracer.cpp:59-71): one frame; returns the accumulated image,
		// fills *textureDebug.  The seed is the fixed sequence seedStep * (n + 1) instead of the
		// wall clock (:63,78-82) so that renders are reproducible.
		std::vector<float> generateImage( std::vector<float>* textureDebug );
		// `frames` x generateImage in one device launch (no per-frame host round trip).
		std::vector<float> generateImages( uint32_t frames );

		void resetSampleCount();
		void setCamera( Camera* camera ) { mCamera = camera; }
		void setFocus( int x, int y );
		void setFOV( float fov ) { mFOV = fov; }
		void setWidthAndHeight( uint32_t width, uint32_t height ) { mWidth = width; mHeight = height; }
		void setSeedStep( float step ) { mSeedStep = step; }
		void setTiles( uint32_t world, uint32_t rank ) { mTileWorld = world; mTileRank = rank; }

		uint32_t getSampleCount() const { return mSampleCount; }
		double getKernelTime() const;
		pbr_ctx* context() { return mCtx; }
		const SceneBuffers& buffers() const { return mBuffers; }
		const pbr_camera& camera() const { return mStructCam; }

		// initKernelArgs' pixel size (PathTracer.cpp:89-91)
		static float pixelDimension( uint32_t width, uint32_t height, float fovDegrees );
		// updateEyeBuffer (PathTracer.cpp:628-652)
		static void fillCameraBasis( const Camera& cam, pbr_camera* out );
		// The pbr_config the current Cfg + buffers imply (CL::setValues, CL.cpp:626-705)
		static pbr_config makeConfig( const SceneBuffers& buffers, uint32_t width, uint32_t height );

	private:
		void updateEyeBuffer();
		void check( int status, const char* what );

		int mDevice;
		uint32_t mWidth, mHeight;
		float mFOV;
		uint32_t mSampleCount;
		float mSeedStep;
		uint32_t mTileWorld, mTileRank;
		pbr_camera mStructCam;
		Camera* mCamera;
		pbr_ctx* mCtx;
		SceneBuffers mBuffers;

};

}  // namespace pbr
