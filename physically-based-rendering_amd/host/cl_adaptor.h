// `CL` look-alike over the HIP core.
//
// The reference reaches its device code only through class CL (source/CL.h:20-83) as driven by
// PathTracer (source/PathTracer.cpp:43-71,88-125,210-230,334-530).  This class has the same public
// methods and the same cl_* types in their signatures, so the reference's PathTracer.cpp compiles
// against it unchanged — but there is no OpenCL underneath: buffers are host blobs until the first
// execute(), the `#NAME#` source substitutions become a pbr_config, the kernel arguments become a
// pbr_scene_desc + pbr_render_frame() (include/pbr_hip.h).
//
//   reference call (PathTracer.cpp)                         what happens here
//   createBuffer( vector, bytes )          :334-507         bytes copied into a host blob; handle returned
//   setReplacement( "#BVH_NUM_NODES#" ...) :210,338,472     kept; #SKY_LIGHT# is parsed back into 4 floats
//   createImage2DReadOnly / WriteOnly      :528-530         image handles (in / out / debug by kernel-arg slot)
//   loadProgram( path ), createKernel      :225-226         configuration read from Cfg as CL::setValues does
//   setKernelArg( k, i, size, data )       :46-48,100-124   argument i recorded (0 seed, 1 weight, 2 pxDim,
//                                                           3 camera, 4 bvh, 5-10 arrays, 11-13 images)
//   execute( k ), finish()                 :50-51           first time: pbr_upload_scene + pbr_configure;
//                                                           every time: pbr_render_frame
//   readImageOutput( image, w, h, dst )    :66-67           pbr_read_output / pbr_read_debug
//   updateImageReadOnly( image, ..., src ) :61              if src holds what readImageOutput just returned:
//                                                           pbr_accumulate (device-side), else pbr_write_input
//
// Errors: the reference logs and carries on (CL.cpp:89-99); here the first failing call throws
// std::runtime_error with pbr_last_error's text.
#pragma once

#include <CL/cl.h>

#include <cstddef>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "pbr_hip.h"

class CL {

	public:
		explicit CL( const bool silent = false );
		~CL();

		template<typename T> cl_mem createBuffer( std::vector<T> object, size_t objectSize ) {
			return this->adoptBytes( object.empty() ? nullptr : (const void*) &object[0], objectSize );
		}

		cl_mem createEmptyBuffer( size_t size, cl_mem_flags flags );
		cl_mem createImage2DReadOnly( size_t width, size_t height, cl_float* data );
		cl_mem createImage2DWriteOnly( size_t width, size_t height );
		cl_kernel createKernel( const char* functionName );
		void execute( cl_kernel kernel );
		void finish();
		void freeBuffers();
		std::map<cl_kernel, std::string> getKernelNames();
		std::map<cl_kernel, double> getKernelTimes();
		void loadProgram( std::string filepath );
		void readImageOutput( cl_mem image, size_t width, size_t height, cl_float* outputTarget );
		void setKernelArg( cl_kernel kernel, cl_uint index, size_t size, void* data );
		void setReplacement( std::string before, std::string after );
		cl_mem updateBuffer( cl_mem buffer, size_t size, void* data );
		cl_mem updateImageReadOnly( cl_mem image, size_t width, size_t height, cl_float* data );

		// not in the reference: the context underneath (tests, multi-frame renders)
		pbr_ctx* context() { return mCtx; }

	private:
		struct Blob {                       // what a cl_mem handle points at
			std::vector<unsigned char> bytes;
			bool isImage = false;
			bool writeOnly = false;
			size_t width = 0, height = 0;
		};

		cl_mem adoptBytes( const void* data, size_t size );
		Blob* blobOf( cl_mem handle );
		const Blob* argBlob( cl_uint index );
		void uploadScene();
		void check( int status, const char* what );

		bool mSilent;
		cl_uint mWorkWidth, mWorkHeight;    // from Cfg at construction, as CL.cpp:18-19
		pbr_ctx* mCtx;
		std::vector<std::unique_ptr<Blob>> mBlobs;
		std::map<std::string, std::string> mReplacements;
		std::map<cl_uint, std::vector<unsigned char>> mArgs;   // kernel-argument slot -> raw bytes
		int mKernelTag;                     // createKernel hands out the address of this as the cl_kernel
		bool mProgramLoaded, mSceneDirty, mInputDirty;
		const cl_float* mLastReadTarget;    // where readImageOutput( imageOut ) last copied to
		std::vector<float> mLastRead;       // and what
		bool mOutputIsFresh = false;        // set by readImageOutput( imageOut ), cleared by the swap it allows and by execute()

};
