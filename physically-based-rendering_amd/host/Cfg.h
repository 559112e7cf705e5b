// Configuration store with the reference's key set and defaults.
// Mirrors Cfg::get().value<T>( key ) (source/Cfg.h:11-18, keys source/Cfg.cpp:4-39,
// defaults config.json).  The reference keeps a boost::property_tree; this is a flat
// string map with typed access and a small JSON reader that accepts `//` comments.
#pragma once

#include <map>
#include <sstream>
#include <string>

namespace pbr {

class Cfg {

	public:
		static Cfg& get();

		// Reads a JSON file (objects, numbers, strings, booleans, // comments); nested keys are
		// joined with '.', as boost::property_tree paths are.  Returns false if unreadable.
		bool loadConfigFile( const char* filepath );
		// Back to config.json's shipped values.
		void resetDefaults();

		template<typename T> T value( const char* key ) const {
			std::map<std::string, std::string>::const_iterator it = mValues.find( key );
			T out = T();

			if( it != mValues.end() ) {
				convert( it->second, &out );
			}

			return out;
		}

		template<typename T> void value( const char* key, T newValue ) {
			std::ostringstream os;
			os.precision( 9 );
			os << newValue;
			mValues[key] = os.str();
		}

		static const char* ACCEL_STRUCT;
		static const char* BVH_MAXFACES;
		static const char* BVH_SAHFACESLIMIT;
		static const char* BVH_SKIPAHEAD;
		static const char* BVH_SKIPAHEAD_CMP;
		static const char* CAM_CENTER_X;
		static const char* CAM_CENTER_Y;
		static const char* CAM_CENTER_Z;
		static const char* CAM_EYE_X;
		static const char* CAM_EYE_Y;
		static const char* CAM_EYE_Z;
		static const char* CAM_LENSE_APERTURE;
		static const char* CAM_LENSE_FOCALLENGTH;
		static const char* PERS_FOV;
		static const char* RENDER_ANTIALIAS;
		static const char* RENDER_BRDF;
		static const char* RENDER_MAXADDEDDEPTH;
		static const char* RENDER_MAXDEPTH;
		static const char* RENDER_PHONGTESS;
		// Not reference keys: the two opt-in modes of the HIP core (include/pbr_hip.h, pbr_config.traversal / .arith).  A viewer that
		// keeps the reference's config.json can switch them there: "hip": { "traversal": 2, "arith": 1 }; absent = 0 = the reference's behaviour.
		static const char* HIP_TRAVERSAL;
		static const char* HIP_ARITH;
		static const char* RENDER_SAMPLES;
		static const char* RENDER_SHADOWRAYS;
		static const char* WINDOW_HEIGHT;
		static const char* WINDOW_WIDTH;

	private:
		Cfg();

		template<typename T> static void convert( const std::string& s, T* out ) {
			std::istringstream is( s );
			is >> *out;
		}

		static void convert( const std::string& s, bool* out ) {
			*out = ( s == "true" || s == "1" );
		}

		static void convert( const std::string& s, std::string* out ) {
			*out = s;
		}

		std::map<std::string, std::string> mValues;

};

}  // namespace pbr
