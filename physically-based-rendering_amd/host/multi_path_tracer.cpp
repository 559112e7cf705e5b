// libpbrmulti.so — include/pbr_multi.h: N contexts of the HIP core in one process, one host thread per device, tile
// sharding, one RCCL all-gather per render.  Build: g++ -std=c++17 -D__HIP_PLATFORM_AMD__ ... -lpbrhip -lrccl -lamdhip64.
#include "multi_path_tracer.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "pbr_hip_diag.h"

namespace pbr {

namespace {

double nowMs() {
	return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count();
}

std::string hipWhat( const char* call, hipError_t err ) {
	return std::string( call ) + ": " + hipGetErrorString( err );
}

}  // namespace


// ---- RankWorker ---------------------------------------------------------------------------------------------------

RankWorker::RankWorker() : mThread( [this] { loop(); } ) {}

RankWorker::~RankWorker() {
	{
		std::lock_guard<std::mutex> lock( mMutex );
		mQuit = true;
	}

	mWake.notify_all();
	mThread.join();
}

void RankWorker::run( std::function<void()> job ) {
	{
		std::unique_lock<std::mutex> lock( mMutex );
		mDone.wait( lock, [this] { return !mBusy; } );
		mJob = std::move( job );
		mBusy = true;
	}

	mWake.notify_all();
}

void RankWorker::wait() {
	std::unique_lock<std::mutex> lock( mMutex );
	mDone.wait( lock, [this] { return !mBusy; } );
}

void RankWorker::loop() {
	for( ;; ) {
		std::function<void()> job;
		{
			std::unique_lock<std::mutex> lock( mMutex );
			mWake.wait( lock, [this] { return mQuit || ( mBusy && mJob ); } );

			if( mQuit ) {
				return;
			}

			job = std::move( mJob );
			mJob = nullptr;
		}

		job();
		{
			std::lock_guard<std::mutex> lock( mMutex );
			mBusy = false;
		}

		mDone.notify_all();
	}
}


void RankBarrier::arrive() {
	std::unique_lock<std::mutex> lock( mMutex );
	const uint64_t generation = mGeneration;

	if( ++mWaiting == mCount ) {
		mWaiting = 0;
		mGeneration++;
		mAll.notify_all();
		return;
	}

	mAll.wait( lock, [&] { return mGeneration != generation; } );
}


// ---- MultiPathTracer ------------------------------------------------------------------------------------------------

MultiPathTracer::MultiPathTracer( const std::vector<int>& devices, int transport )
	: mRanks( devices.size() ), mBarrier( (int) devices.size() ), mTransport( transport ) {
	if( devices.empty() ) {
		throw std::runtime_error( "pbr_multi_create: no devices" );
	}
	if( transport != PBR_MULTI_RCCL && transport != PBR_MULTI_PEER_COPY ) {
		throw std::runtime_error( "pbr_multi_create: transport must be PBR_MULTI_RCCL or PBR_MULTI_PEER_COPY" );
	}
	if( transport == PBR_MULTI_RCCL ) {
		std::vector<int> sorted( devices );
		std::sort( sorted.begin(), sorted.end() );

		if( std::adjacent_find( sorted.begin(), sorted.end() ) != sorted.end() ) {
			throw std::runtime_error( "pbr_multi_create: an RCCL communicator needs distinct devices (PBR_MULTI_PEER_COPY rehearses several ranks on one)" );
		}
	}

	for( size_t r = 0; r < devices.size(); r++ ) {
		mRanks[r].device = devices[r];
	}

	// contexts and streams, every rank on its own thread (a context's device state is set up where it will be used)
	const int made = onEveryRank( [this]( int r ) {
		Rank& rank = mRanks[(size_t) r];
		const int status = pbr_create( rank.device, &rank.ctx );

		if( status != PBR_OK ) {
			return failed( r, status, rank.ctx != nullptr ? pbr_last_error( rank.ctx ) : "pbr_create failed (no HIP device?)" );
		}

		hipError_t err = hipSetDevice( rank.device );

		if( err == hipSuccess ) {
			err = hipStreamCreateWithFlags( &rank.stream, hipStreamNonBlocking );
		}

		return ( err == hipSuccess ) ? PBR_OK : failed( r, PBR_EDEVICE, hipWhat( "hipStreamCreate", err ) );
	} );

	if( made != PBR_OK ) {
		const std::string why = mError;
		release();
		throw std::runtime_error( why );
	}

	if( transport == PBR_MULTI_RCCL ) {
		std::vector<ncclComm_t> comms( devices.size(), nullptr );
		const ncclResult_t res = ncclCommInitAll( comms.data(), (int) devices.size(), devices.data() );

		if( res != ncclSuccess ) {
			const std::string why = std::string( "ncclCommInitAll: " ) + ncclGetErrorString( res );
			release();
			throw std::runtime_error( why );
		}

		for( size_t r = 0; r < devices.size(); r++ ) {
			mRanks[r].comm = comms[r];
		}
	}
}

MultiPathTracer::~MultiPathTracer() {
	release();
}

void MultiPathTracer::release() {
	for( Rank& rank : mRanks ) {
		rank.worker.wait();
	}

	freeBuffers();

	for( Rank& rank : mRanks ) {
		if( rank.comm != nullptr ) {
			(void) ncclCommDestroy( rank.comm );
			rank.comm = nullptr;
		}
		if( rank.stream != nullptr ) {
			(void) hipSetDevice( rank.device );
			(void) hipStreamDestroy( rank.stream );
			rank.stream = nullptr;
		}
		if( rank.ctx != nullptr ) {
			pbr_destroy( rank.ctx );
			rank.ctx = nullptr;
		}
	}
}

void MultiPathTracer::freeBuffers() {
	for( Rank& rank : mRanks ) {
		if( rank.dSend != nullptr || rank.dRecv != nullptr ) {
			(void) hipSetDevice( rank.device );
			(void) hipFree( rank.dSend );
			(void) hipFree( rank.dRecv );
			rank.dSend = rank.dRecv = nullptr;
		}
	}

	mConfigured = false;
}

int MultiPathTracer::failed( int rank, int status, const std::string& what ) {
	Rank& r = mRanks[(size_t) rank];
	r.status = status;
	char head[48];
	std::snprintf( head, sizeof( head ), "rank %d (device %d): ", rank, r.device );
	r.message = head + what;
	return status;
}

int MultiPathTracer::onEveryRank( const std::function<int( int )>& job ) {
	for( size_t r = 0; r < mRanks.size(); r++ ) {
		mRanks[r].status = PBR_OK;
		mRanks[r].message.clear();
		mRanks[r].worker.run( [this, r, &job] {
			const int status = job( (int) r );

			if( status != PBR_OK && mRanks[r].status == PBR_OK ) {
				failed( (int) r, status, mRanks[r].ctx != nullptr ? pbr_last_error( mRanks[r].ctx ) : "failed" );
			}
		} );
	}

	// the first rank that failed ON ITS OWN: a rank that only skipped the exchange because another one had failed reports
	// that, and is passed over as long as the one that caused it is there to name
	int first = PBR_OK;
	bool firstIsConsequence = false;

	for( size_t r = 0; r < mRanks.size(); r++ ) {
		mRanks[r].worker.wait();
		const bool consequence = ( mRanks[r].message.find( "another rank failed" ) != std::string::npos );

		if( mRanks[r].status != PBR_OK && ( first == PBR_OK || ( firstIsConsequence && !consequence ) ) ) {
			first = mRanks[r].status;
			mError = mRanks[r].message;
			firstIsConsequence = consequence;
		}
	}

	return first;
}

int MultiPathTracer::uploadScene( const pbr_scene_desc* scene ) {
	return onEveryRank( [&]( int r ) { return pbr_upload_scene( mRanks[(size_t) r].ctx, scene ); } );
}

int MultiPathTracer::configure( const pbr_config* cfg ) {
	if( cfg == nullptr ) {
		mError = "pbr_multi_configure: null config";
		return PBR_EINVAL;
	}

	freeBuffers();
	const int status = onEveryRank( [&]( int r ) {
		Rank& rank = mRanks[(size_t) r];
		pbr_config mine = *cfg;
		mine.tile_world = (uint32_t) mRanks.size();
		mine.tile_rank = (uint32_t) r;
		const int configured = pbr_configure( rank.ctx, &mine );

		if( configured != PBR_OK ) {
			return configured;
		}

		const uint64_t bytes = pbr_tile_bytes( rank.ctx );
		hipError_t err = hipSetDevice( rank.device );

		if( err == hipSuccess ) {
			err = hipMalloc( &rank.dSend, bytes );
		}
		if( err == hipSuccess ) {
			err = hipMalloc( &rank.dRecv, bytes * mRanks.size() );
		}

		return ( err == hipSuccess ) ? PBR_OK : failed( r, PBR_EDEVICE, hipWhat( "hipMalloc (exchange buffers)", err ) );
	} );

	if( status == PBR_OK ) {
		mTileBytes = pbr_tile_bytes( mRanks[0].ctx );     // the padded size: the same on every rank
		mConfigured = true;
	}

	return status;
}

int MultiPathTracer::resetAccum() {
	return onEveryRank( [&]( int r ) { return pbr_reset_accum( mRanks[(size_t) r].ctx ); } );
}

int MultiPathTracer::electPlan( const std::vector<int>& votes ) {
	int best = -1, bestCount = 0;

	for( size_t i = 0; i < votes.size(); i++ ) {
		if( votes[i] < 0 ) {
			continue;
		}

		const int count = (int) std::count( votes.begin(), votes.end(), votes[i] );

		if( count > bestCount ) {     // strictly more: among equals the first (lowest rank's) vote stays
			best = votes[i];
			bestCount = count;
		}
	}

	return best;
}

int MultiPathTracer::tune( uint32_t framesPerCall, float pxDim, const pbr_camera* cam, int* plan, int* votesOut ) {
	if( !mConfigured || cam == nullptr || framesPerCall == 0 ) {
		mError = "pbr_multi_tune: configure first; camera and frames_per_call must be given";
		return PBR_EINVAL;
	}

	std::vector<int> votes( mRanks.size(), -1 );
	const int status = onEveryRank( [&]( int r ) {
		pbr_ctx* ctx = mRanks[(size_t) r].ctx;
		int pinned = pbr_diag_pin_plan( ctx, -1 );      // the tuner chooses

		if( pinned != PBR_OK ) {
			return pinned;
		}

		uint32_t budget = 0;
		int status = pbr_diag_tune_budget( ctx, &budget );
		std::vector<float> seeds( framesPerCall );
		uint32_t done = 0;
		char name[48];
		int tuned = -1;

		while( status == PBR_OK && ( done < budget || tuned < 0 ) && done < 4u * std::max<uint32_t>( budget, 1u ) ) {
			for( uint32_t k = 0; k < framesPerCall; k++ ) {
				seeds[k] = 0.0333f * (float) ( done + k + 1u );     // the fixed sequence that stands in for the wall clock (PathTracer.cpp:63)
			}

			status = pbr_render( ctx, done, framesPerCall, seeds.data(), pxDim, cam );
			done += framesPerCall;

			if( status == PBR_OK ) {
				status = pbr_diag_last_plan( ctx, name, sizeof( name ), &tuned );
			}
		}

		votes[(size_t) r] = tuned;
		return ( status == PBR_OK ) ? pbr_reset_accum( ctx ) : status;
	} );

	if( status != PBR_OK ) {
		return status;
	}

	const int elected = electPlan( votes );

	if( plan != nullptr ) {
		*plan = elected;
	}
	if( votesOut != nullptr ) {
		std::copy( votes.begin(), votes.end(), votesOut );
	}

	return onEveryRank( [&]( int r ) { return pbr_diag_pin_plan( mRanks[(size_t) r].ctx, elected ); } );
}

// One rank's share of the exchange, on its own thread: its tiles into its send buffer, the all-gather, the scatter into
// its full frame.  RCCL: every rank's thread calls ncclAllGather on its own communicator and stream — the one-thread-per-
// device form of a single-process communicator (rccl.h: collective calls of one process "must be called by different
// threads / processes or use ncclGroupStart / ncclGroupEnd").
int MultiPathTracer::exchange( int r ) {
	Rank& rank = mRanks[(size_t) r];
	const bool peer = ( mTransport == PBR_MULTI_PEER_COPY );
	const double t0 = nowMs();
	int status = pbr_export_tiles( rank.ctx, rank.dSend );
	hipError_t err = ( status == PBR_OK ) ? hipSetDevice( rank.device ) : hipSuccess;

	// First meeting point, both transports: every send buffer is written (pbr_export_tiles has waited for its copy) — and every rank
	// knows whether ALL ranks got this far.  A collective that one rank never enters would hang the others; a rank that has failed
	// (its render or its export) says so here, and then nobody enters it.
	if( status != PBR_OK || err != hipSuccess ) {
		mFailedRanks.fetch_add( 1 );
	}

	mBarrier.arrive();
	const bool everyoneHere = ( mFailedRanks.load() == 0 );

	if( status == PBR_OK && err == hipSuccess && everyoneHere ) {
		if( peer ) {
			for( size_t other = 0; other < mRanks.size() && err == hipSuccess; other++ ) {
				err = hipMemcpyPeerAsync( (char*) rank.dRecv + other * mTileBytes, rank.device, mRanks[other].dSend, mRanks[other].device, mTileBytes, rank.stream );
			}
		}
		else {
			const ncclResult_t res = ncclAllGather( rank.dSend, rank.dRecv, (size_t) ( mTileBytes / sizeof( float ) ), ncclFloat, rank.comm, rank.stream );

			if( res != ncclSuccess ) {
				status = failed( r, PBR_EDEVICE, std::string( "ncclAllGather: " ) + ncclGetErrorString( res ) );
			}
		}

		if( status == PBR_OK && err == hipSuccess ) {
			err = hipStreamSynchronize( rank.stream );
		}
	}

	// Second meeting point: nobody overwrites a send buffer (the next export) while somebody still reads it; the failure count is
	// reset for the next exchange by the last rank to pass.
	mBarrier.arrive();

	if( r == 0 ) {
		mFailedRanks.store( 0 );
	}

	mBarrier.arrive();      // ... and nobody starts the next exchange before it is

	if( status != PBR_OK ) {
		return status;
	}
	if( err != hipSuccess ) {
		return failed( r, PBR_EDEVICE, hipWhat( "tile exchange", err ) );
	}
	if( !everyoneHere ) {
		return failed( r, PBR_ESTATE, "the tile exchange was skipped: another rank failed before it" );
	}

	status = pbr_import_tiles( rank.ctx, rank.dRecv );
	rank.gatherMs = nowMs() - t0;
	return status;
}

// A rank whose render failed does not exchange, but the others must not wait for it: it passes the meeting points as a failed rank.
void MultiPathTracer::skipExchange() {
	mFailedRanks.fetch_add( 1 );
	mBarrier.arrive();
	mBarrier.arrive();
	mBarrier.arrive();
}

int MultiPathTracer::gather() {
	if( !mConfigured ) {
		mError = "pbr_multi_gather before pbr_multi_configure";
		return PBR_ESTATE;
	}

	return onEveryRank( [&]( int r ) { return exchange( r ); } );
}

int MultiPathTracer::render( uint32_t firstSampleCount, uint32_t nFrames, const float* seeds, float pxDim, const pbr_camera* cam, bool withGather ) {
	if( !mConfigured ) {
		mError = "pbr_multi_render before pbr_multi_configure";
		return PBR_ESTATE;
	}

	return onEveryRank( [&]( int r ) {
		Rank& rank = mRanks[(size_t) r];
		const double t0 = nowMs();
		const int status = pbr_render( rank.ctx, firstSampleCount, nFrames, seeds, pxDim, cam );
		rank.renderMs = nowMs() - t0;
		rank.gatherMs = 0.0;

		if( status != PBR_OK ) {
			if( withGather ) {
				skipExchange();
			}

			return status;
		}

		return withGather ? exchange( r ) : PBR_OK;
	} );
}

int MultiPathTracer::renderFrame( float seed, float pixelWeight, float pxDim, const pbr_camera* cam, bool accumulate, bool withGather ) {
	if( !mConfigured || cam == nullptr ) {
		mError = "pbr_multi_render_frame: configure first; the camera must be given";
		return PBR_ESTATE;
	}

	// depth of field: every pixel reads the previous-frame distance of the focus pixel (pathtracing.cl:58-65), whose tile
	// lives on ONE rank.  In one process the hand-over is a host float, not an ncclBroadcast.
	if( cam->focusPoint[0] >= 0 && cam->focusPoint[1] >= 0 && mRanks.size() > 1 ) {
		float depth = 0.0f;
		bool found = false;

		for( Rank& rank : mRanks ) {
			float t = 0.0f;
			int owned = 0;
			const int status = pbr_get_focus_depth( rank.ctx, cam->focusPoint[0], cam->focusPoint[1], &t, &owned );

			if( status != PBR_OK ) {
				mError = pbr_last_error( rank.ctx );
				return status;
			}
			if( owned != 0 ) {
				depth = t;
				found = true;
			}
		}

		if( !found ) {
			mError = "pbr_multi_render_frame: no rank owns the focus pixel";
			return PBR_EINVAL;
		}

		for( Rank& rank : mRanks ) {
			const int status = pbr_set_focus_depth( rank.ctx, depth );

			if( status != PBR_OK ) {
				mError = pbr_last_error( rank.ctx );
				return status;
			}
		}
	}

	return onEveryRank( [&]( int r ) {
		Rank& rank = mRanks[(size_t) r];
		const double t0 = nowMs();
		int status = pbr_render_frame( rank.ctx, seed, pixelWeight, pxDim, cam );
		rank.renderMs = nowMs() - t0;
		rank.gatherMs = 0.0;

		if( status == PBR_OK && withGather ) {
			status = exchange( r );
		}
		else if( withGather ) {
			skipExchange();
		}

		if( status == PBR_OK && accumulate ) {
			status = pbr_accumulate( rank.ctx );
		}

		return status;
	} );
}

int MultiPathTracer::readFull( int rank, float* rgba ) {
	if( rank < 0 || rank >= size() ) {
		mError = "pbr_multi_read_full: no such rank";
		return PBR_EINVAL;
	}

	const int status = pbr_read_full( mRanks[(size_t) rank].ctx, rgba );

	if( status != PBR_OK ) {
		mError = pbr_last_error( mRanks[(size_t) rank].ctx );
	}

	return status;
}

void MultiPathTracer::timings( double* renderMs, double* gatherMs ) const {
	for( size_t r = 0; r < mRanks.size(); r++ ) {
		if( renderMs != nullptr ) {
			renderMs[r] = mRanks[r].renderMs;
		}
		if( gatherMs != nullptr ) {
			gatherMs[r] = mRanks[r].gatherMs;
		}
	}
}

}  // namespace pbr


// ---- the C ABI (include/pbr_multi.h) ----------------------------------------------------------------------------------

struct pbr_multi {
	pbr::MultiPathTracer* impl = nullptr;
	std::string error;
};

namespace {

thread_local std::string gCreateError;

int done( pbr_multi* m, int status ) {
	if( status != PBR_OK ) {
		m->error = m->impl->lastError();
	}

	return status;
}

}  // namespace

extern "C" {

int pbr_multi_create( const int* devices, int count, int transport, pbr_multi** out ) {
	if( out == nullptr ) {
		return PBR_EINVAL;
	}

	*out = nullptr;

	if( devices == nullptr || count <= 0 ) {
		gCreateError = "pbr_multi_create: no devices";
		return PBR_EINVAL;
	}

	pbr_multi* m = new pbr_multi();

	try {
		m->impl = new pbr::MultiPathTracer( std::vector<int>( devices, devices + count ), transport );
	}
	catch( const std::exception& e ) {
		gCreateError = e.what();
		delete m;
		return PBR_EDEVICE;
	}

	*out = m;
	return PBR_OK;
}

void pbr_multi_destroy( pbr_multi* m ) {
	if( m != nullptr ) {
		delete m->impl;
		delete m;
	}
}

const char* pbr_multi_last_error( const pbr_multi* m ) {
	return ( m != nullptr ) ? m->error.c_str() : gCreateError.c_str();
}

int pbr_multi_size( const pbr_multi* m ) {
	return ( m != nullptr ) ? m->impl->size() : 0;
}

pbr_ctx* pbr_multi_context( pbr_multi* m, int rank ) {
	return ( m != nullptr && rank >= 0 && rank < m->impl->size() ) ? m->impl->context( rank ) : nullptr;
}

int pbr_multi_upload_scene( pbr_multi* m, const pbr_scene_desc* scene ) {
	return ( m == nullptr ) ? PBR_EINVAL : done( m, m->impl->uploadScene( scene ) );
}

int pbr_multi_configure( pbr_multi* m, const pbr_config* cfg ) {
	return ( m == nullptr ) ? PBR_EINVAL : done( m, m->impl->configure( cfg ) );
}

int pbr_multi_reset_accum( pbr_multi* m ) {
	return ( m == nullptr ) ? PBR_EINVAL : done( m, m->impl->resetAccum() );
}

int pbr_multi_tune( pbr_multi* m, uint32_t frames_per_call, float pxDim, const pbr_camera* cam, int* plan, int* votes ) {
	return ( m == nullptr ) ? PBR_EINVAL : done( m, m->impl->tune( frames_per_call, pxDim, cam, plan, votes ) );
}

int pbr_multi_render( pbr_multi* m, uint32_t first_sample_count, uint32_t n_frames, const float* seeds, float pxDim, const pbr_camera* cam, int gather ) {
	return ( m == nullptr ) ? PBR_EINVAL : done( m, m->impl->render( first_sample_count, n_frames, seeds, pxDim, cam, gather != 0 ) );
}

int pbr_multi_render_frame( pbr_multi* m, float seed, float pixelWeight, float pxDim, const pbr_camera* cam, int accumulate, int gather ) {
	return ( m == nullptr ) ? PBR_EINVAL : done( m, m->impl->renderFrame( seed, pixelWeight, pxDim, cam, accumulate != 0, gather != 0 ) );
}

int pbr_multi_gather( pbr_multi* m ) {
	return ( m == nullptr ) ? PBR_EINVAL : done( m, m->impl->gather() );
}

int pbr_multi_read_full( pbr_multi* m, int rank, float* rgba ) {
	return ( m == nullptr ) ? PBR_EINVAL : done( m, m->impl->readFull( rank, rgba ) );
}

int pbr_multi_timings( const pbr_multi* m, double* render_ms, double* gather_ms ) {
	if( m == nullptr ) {
		return PBR_EINVAL;
	}

	m->impl->timings( render_ms, gather_ms );
	return PBR_OK;
}

}  // extern "C"
