// The path on N GPUs of one node from one process: N contexts of the HIP core, one host thread each, tile sharding and
// one RCCL all-gather per render.  C++ face of include/pbr_multi.h (which documents the design); the reference has no
// counterpart (one CL* per PathTracer on one device: source/PathTracer.cpp:150-153, source/CL.cpp:355,521).
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "pbr_hip.h"
#include "pbr_multi.h"

namespace pbr {

// One host thread that runs the jobs it is handed, in order (a context's calls must come from one thread at a time, and a
// frame-by-frame caller should not pay for N thread starts per frame).
class RankWorker {

	public:
		RankWorker();
		~RankWorker();
		void run( std::function<void()> job );   // returns at once
		void wait();                             // until the job has finished

	private:
		void loop();

		std::thread mThread;
		std::mutex mMutex;
		std::condition_variable mWake, mDone;
		std::function<void()> mJob;
		bool mBusy = false, mQuit = false;

};


// All ranks meet here (the peer-copy exchange reads the other ranks' buffers only after every one is written).
class RankBarrier {

	public:
		explicit RankBarrier( int count ) : mCount( count ) {}
		void arrive();

	private:
		std::mutex mMutex;
		std::condition_variable mAll;
		int mCount, mWaiting = 0;
		uint64_t mGeneration = 0;

};


class MultiPathTracer {

	public:
		MultiPathTracer( const std::vector<int>& devices, int transport );   // throws std::runtime_error
		~MultiPathTracer();

		int size() const { return (int) mRanks.size(); }
		pbr_ctx* context( int rank ) { return mRanks[(size_t) rank].ctx; }
		const std::string& lastError() const { return mError; }

		int uploadScene( const pbr_scene_desc* scene );
		int configure( const pbr_config* cfg );
		int resetAccum();
		int tune( uint32_t framesPerCall, float pxDim, const pbr_camera* cam, int* plan, int* votes );
		int render( uint32_t firstSampleCount, uint32_t nFrames, const float* seeds, float pxDim, const pbr_camera* cam, bool gather );
		int renderFrame( float seed, float pixelWeight, float pxDim, const pbr_camera* cam, bool accumulate, bool gather );
		int gather();
		int readFull( int rank, float* rgba );
		void timings( double* renderMs, double* gatherMs ) const;

		// the plan most ranks voted for (-1 = no vote); ties go to the lowest rank's vote — bench.py's elect_plan
		static int electPlan( const std::vector<int>& votes );

	private:
		struct Rank {
			int device = 0;
			pbr_ctx* ctx = nullptr;
			hipStream_t stream = nullptr;
			void* dSend = nullptr;      // this rank's compact tile buffer (pbr_tile_bytes)
			void* dRecv = nullptr;      // size() of them: the all-gather layout pbr_import_tiles reads
			ncclComm_t comm = nullptr;
			RankWorker worker;
			int status = 0;
			std::string message;
			double renderMs = 0.0, gatherMs = 0.0;
		};

		// `job( rank )` on every rank's thread at the same time; the first failing rank's status (its message in mError)
		int onEveryRank( const std::function<int( int )>& job );
		int exchange( int rank );
		void skipExchange();
		int failed( int rank, int status, const std::string& what );
		void freeBuffers();
		void release();

		std::vector<Rank> mRanks;
		RankBarrier mBarrier;
		std::atomic<int> mFailedRanks{ 0 };   // of the exchange in progress (exchange / skipExchange)
		int mTransport;
		uint64_t mTileBytes = 0;
		bool mConfigured = false;
		std::string mError;

};

}  // namespace pbr
