#include "Cfg.h"

#include <cctype>
#include <fstream>
#include <iterator>

namespace pbr {

const char* Cfg::ACCEL_STRUCT = "accel_struct";
const char* Cfg::BVH_MAXFACES = "bvh.max_faces";
const char* Cfg::BVH_SAHFACESLIMIT = "bvh.sah_faces_limit";
const char* Cfg::BVH_SKIPAHEAD = "bvh.skip_ahead";
const char* Cfg::BVH_SKIPAHEAD_CMP = "bvh.skip_ahead_compare";
const char* Cfg::CAM_CENTER_X = "camera.center.x";
const char* Cfg::CAM_CENTER_Y = "camera.center.y";
const char* Cfg::CAM_CENTER_Z = "camera.center.z";
const char* Cfg::CAM_EYE_X = "camera.eye.x";
const char* Cfg::CAM_EYE_Y = "camera.eye.y";
const char* Cfg::CAM_EYE_Z = "camera.eye.z";
const char* Cfg::CAM_LENSE_APERTURE = "camera.thin_lense.aperture";
const char* Cfg::CAM_LENSE_FOCALLENGTH = "camera.thin_lense.focal_length";
const char* Cfg::PERS_FOV = "camera.perspective.fov";
const char* Cfg::RENDER_ANTIALIAS = "render.antialiasing";
const char* Cfg::RENDER_BRDF = "render.brdf";
const char* Cfg::RENDER_MAXADDEDDEPTH = "render.max_added_depth";
const char* Cfg::RENDER_MAXDEPTH = "render.max_depth";
const char* Cfg::RENDER_PHONGTESS = "render.phong_tessellation";
const char* Cfg::HIP_TRAVERSAL = "hip.traversal";
const char* Cfg::HIP_ARITH = "hip.arith";
const char* Cfg::RENDER_SAMPLES = "render.samples";
const char* Cfg::RENDER_SHADOWRAYS = "render.shadow_rays";
const char* Cfg::WINDOW_HEIGHT = "window.height";
const char* Cfg::WINDOW_WIDTH = "window.width";


Cfg& Cfg::get() {
	static Cfg instance;
	return instance;
}


Cfg::Cfg() {
	this->resetDefaults();
}


// The values the reference ships in config.json (lines 3-25, 39-58, 91-111, 121-123).
void Cfg::resetDefaults() {
	mValues.clear();
	mValues[ACCEL_STRUCT] = "0";
	mValues[BVH_MAXFACES] = "2";
	mValues[BVH_SAHFACESLIMIT] = "100000";
	mValues[BVH_SKIPAHEAD] = "true";
	mValues[BVH_SKIPAHEAD_CMP] = "0.7";
	mValues[CAM_CENTER_X] = "0.0";
	mValues[CAM_CENTER_Y] = "0.0";
	mValues[CAM_CENTER_Z] = "1.0";
	mValues[CAM_EYE_X] = "0.0";
	mValues[CAM_EYE_Y] = "1.0";
	mValues[CAM_EYE_Z] = "3.0";
	mValues[CAM_LENSE_APERTURE] = "1.8";
	mValues[CAM_LENSE_FOCALLENGTH] = "0.035";
	mValues[PERS_FOV] = "45.0";
	mValues[RENDER_ANTIALIAS] = "0.7";
	mValues[RENDER_BRDF] = "1";
	mValues[RENDER_MAXADDEDDEPTH] = "5";
	mValues[RENDER_MAXDEPTH] = "3";
	mValues[RENDER_PHONGTESS] = "0.0";
	mValues[HIP_TRAVERSAL] = "0";
	mValues[HIP_ARITH] = "0";
	mValues[RENDER_SAMPLES] = "1";
	mValues[RENDER_SHADOWRAYS] = "0";
	mValues[WINDOW_HEIGHT] = "600";
	mValues[WINDOW_WIDTH] = "800";
}


namespace {

struct JsonReader {
	const std::string& s;
	size_t i;
	std::map<std::string, std::string>* out;
	bool ok;

	void skip() {
		while( i < s.size() ) {
			if( std::isspace( (unsigned char) s[i] ) ) {
				i++;
			}
			else if( s[i] == '/' && i + 1 < s.size() && s[i + 1] == '/' ) {
				while( i < s.size() && s[i] != '\n' ) {
					i++;
				}
			}
			else {
				break;
			}
		}
	}

	std::string str() {
		std::string r;
		i++;  // opening quote

		while( i < s.size() && s[i] != '"' ) {
			if( s[i] == '\\' && i + 1 < s.size() ) {
				i++;
			}
			r.push_back( s[i++] );
		}

		i++;  // closing quote
		return r;
	}

	void object( const std::string& prefix ) {
		skip();

		if( i >= s.size() || s[i] != '{' ) {
			ok = false;
			return;
		}

		i++;

		while( ok ) {
			skip();

			if( i >= s.size() ) {
				ok = false;
				return;
			}
			if( s[i] == '}' ) {
				i++;
				return;
			}
			if( s[i] == ',' ) {
				i++;
				continue;
			}
			if( s[i] != '"' ) {
				ok = false;
				return;
			}

			const std::string key = prefix.empty() ? str() : prefix + "." + str();
			skip();

			if( i >= s.size() || s[i] != ':' ) {
				ok = false;
				return;
			}

			i++;
			skip();

			if( i < s.size() && s[i] == '{' ) {
				object( key );
			}
			else if( i < s.size() && s[i] == '"' ) {
				( *out )[key] = str();
			}
			else {
				std::string v;

				while( i < s.size() && s[i] != ',' && s[i] != '}' && !std::isspace( (unsigned char) s[i] ) ) {
					v.push_back( s[i++] );
				}

				( *out )[key] = v;
			}
		}
	}
};

}  // namespace


bool Cfg::loadConfigFile( const char* filepath ) {
	std::ifstream in( filepath );

	if( !in ) {
		return false;
	}

	const std::string text( ( std::istreambuf_iterator<char>( in ) ), std::istreambuf_iterator<char>() );
	JsonReader r = { text, 0, &mValues, true };
	r.object( "" );

	return r.ok;
}

}  // namespace pbr
