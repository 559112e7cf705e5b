#include "model_io.h"

#include <algorithm>
#include <cstdlib>
#include <fstream>

#include "Cfg.h"

namespace pbr {

namespace {

// boost::algorithm::trim
std::string trimmed( const std::string& s ) {
	size_t a = 0, b = s.size();

	while( a < b && std::isspace( (unsigned char) s[a] ) ) {
		a++;
	}
	while( b > a && std::isspace( (unsigned char) s[b - 1] ) ) {
		b--;
	}

	return s.substr( a, b - a );
}

// boost::split( parts, line, boost::is_any_of( seps ) ) with token compression OFF:
// adjacent separators yield empty tokens, exactly as the reference's parsers see them.
std::vector<std::string> tokens( const std::string& s, const char* seps ) {
	std::vector<std::string> parts;
	std::string cur;

	for( size_t i = 0; i < s.size(); i++ ) {
		bool isSep = false;

		for( const char* p = seps; *p; p++ ) {
			isSep = isSep || ( s[i] == *p );
		}

		if( isSep ) {
			parts.push_back( cur );
			cur.clear();
		}
		else {
			cur.push_back( s[i] );
		}
	}

	parts.push_back( cur );

	return parts;
}

float toFloat( const std::string& s ) {
	return (float) atof( s.c_str() );
}

std::string swapExtension( std::string file, const char* ext ) {
	const size_t at = file.rfind( ".obj" );

	if( at != std::string::npos ) {
		file.replace( at, 4, ext );
	}

	return file;
}

}  // namespace


// ---------------------------------------------------------------------------
// MtlParser::load — source/MtlParser.cpp:52-236.  Keys: newmtl d Tr illum Ka Kd Ks Ni Ns
// light rough p nu nv Rs Rd.  `Tr` is ignored once any `d` was seen in the file.
// ---------------------------------------------------------------------------
void MtlParser::load( std::string file ) {
	mMaterials.clear();

	std::ifstream in( file.c_str() );

	if( !in ) {
		return;
	}

	material_t mtl;
	int found = 0;
	bool sawDissolve = false;
	std::string raw;

	while( in.good() ) {
		std::getline( in, raw );
		const std::string line = trimmed( raw );

		if( line.length() < 3 || line[0] == '#' ) {
			continue;
		}

		const std::vector<std::string> parts = tokens( line, " \t" );
		const std::string& key = parts[0];
		const size_t n = parts.size();

		if( key == "newmtl" ) {
			if( n < 2 ) {
				continue;
			}
			if( found > 0 ) {
				mMaterials.push_back( mtl );
			}

			found++;
			mtl = material_t();
			mtl.mtlName = parts[1];
		}
		else if( key == "d" ) {
			if( n >= 2 ) {
				mtl.d = toFloat( parts[1] );
				sawDissolve = true;
			}
		}
		else if( key == "Tr" && !sawDissolve ) {
			if( n >= 2 ) {
				mtl.d = (float) ( 1.0f - atof( parts[1].c_str() ) );
			}
		}
		else if( key == "illum" ) {
			if( n >= 2 ) {
				mtl.illum = (int8_t) atol( parts[1].c_str() );

				if( mtl.illum < 0 || mtl.illum > 10 ) {
					mtl.illum = 2;
				}
			}
		}
		else if( key == "Ka" || key == "Kd" || key == "Ks" ) {
			if( n >= 4 ) {
				float4_t& c = ( key == "Ka" ) ? mtl.Ka : ( key == "Kd" ) ? mtl.Kd : mtl.Ks;
				c.x = toFloat( parts[1] );
				c.y = toFloat( parts[2] );
				c.z = toFloat( parts[3] );
			}
		}
		else if( n >= 2 ) {
			const float v = toFloat( parts[1] );

			if( key == "Ni" ) { mtl.Ni = v; }
			else if( key == "Ns" ) { mtl.Ns = v; }
			else if( key == "light" ) { mtl.light = (int8_t) atoi( parts[1].c_str() ); }
			else if( key == "rough" ) { mtl.rough = v; }
			else if( key == "p" ) { mtl.p = v; }
			else if( key == "nu" ) { mtl.nu = v; }
			else if( key == "nv" ) { mtl.nv = v; }
			else if( key == "Rs" ) { mtl.Rs = v; }
			else if( key == "Rd" ) { mtl.Rd = v; }
		}
	}

	if( found > 0 ) {
		mMaterials.push_back( mtl );
	}
}


// ---------------------------------------------------------------------------
// LightParser::load — source/LightParser.cpp:39-128.  Keys: newlight type rgb pos radius.
// A file without lights switches render.shadow_rays off (LightParser.cpp:119-121).
// ---------------------------------------------------------------------------
void LightParser::load( std::string file ) {
	mLights.clear();

	std::ifstream in( file.c_str() );

	if( !in ) {
		return;
	}

	light_t light;
	int found = 0;
	std::string raw;

	while( in.good() ) {
		std::getline( in, raw );
		const std::string line = trimmed( raw );

		if( line.length() < 3 || line[0] == '#' ) {
			continue;
		}

		const std::vector<std::string> parts = tokens( line, " \t" );
		const std::string& key = parts[0];
		const size_t n = parts.size();

		if( key == "newlight" ) {
			if( n < 2 ) {
				continue;
			}
			if( found > 0 ) {
				mLights.push_back( light );
			}

			found++;
			light = light_t();
			light.lightName = parts[1];
		}
		else if( key == "type" && n >= 2 ) {
			light.type = (uint32_t) atol( parts[1].c_str() );
		}
		else if( key == "rgb" && n >= 4 ) {
			light.rgb.x = toFloat( parts[1] );
			light.rgb.y = toFloat( parts[2] );
			light.rgb.z = toFloat( parts[3] );
		}
		else if( key == "pos" && n >= 4 ) {
			light.pos.x = toFloat( parts[1] );
			light.pos.y = toFloat( parts[2] );
			light.pos.z = toFloat( parts[3] );
		}
		else if( key == "radius" && n >= 2 ) {
			light.radius = toFloat( parts[1] );
		}
	}

	if( found > 0 ) {
		mLights.push_back( light );
	}
	else {
		Cfg::get().value( Cfg::RENDER_SHADOWRAYS, 0 );
	}
}


// ---------------------------------------------------------------------------
// ObjParser::parseFace — source/ObjParser.cpp:266-308.  Triangles only.  Each corner is
// split at '/': exactly two fields are read as (v, vn) — so "v/vt" is mis-read as "v//vn" —
// otherwise field 0 is v, field 1 (if any) vt, field 2 (if any) vn; "v//vn" therefore stores
// atol("") - 1 = UINT_MAX in the vt slot.  Indices are 1-based in the file.
// ---------------------------------------------------------------------------
void ObjParser::parseFace(
	const std::string& line, std::vector<uint32_t>* fV, std::vector<uint32_t>* fVN, std::vector<uint32_t>* fVT
) {
	const std::vector<std::string> parts = tokens( line, " \t" );

	for( size_t i = 1; i < parts.size(); i++ ) {
		const std::vector<std::string> e = tokens( parts[i], "/" );

		if( e.size() == 2 ) {
			fV->push_back( (uint32_t) atol( e[0].c_str() ) - 1u );
			fVN->push_back( (uint32_t) atol( e[1].c_str() ) - 1u );
			continue;
		}

		fV->push_back( (uint32_t) atol( e[0].c_str() ) - 1u );

		if( e.size() >= 2 ) {
			fVT->push_back( (uint32_t) atol( e[1].c_str() ) - 1u );
		}
		if( e.size() >= 3 ) {
			fVN->push_back( (uint32_t) atol( e[2].c_str() ) - 1u );
		}
	}
}


// ---------------------------------------------------------------------------
// ObjParser::load — source/ObjParser.cpp:121-221.  `.lights` is read only when
// render.shadow_rays > 0 (:133-135); `.mtl` always (:137).  One material index per `f` line
// (-1 without a preceding usemtl); faces before the first `o` line belong to no object.
// ---------------------------------------------------------------------------
void ObjParser::load( std::string filepath, std::string filename ) {
	mModel = SceneModel();

	const std::string file = filepath.append( filename );
	std::ifstream in( file.c_str() );

	if( Cfg::get().value<int>( Cfg::RENDER_SHADOWRAYS ) > 0 ) {
		LightParser lp;
		lp.load( swapExtension( file, ".lights" ) );
		mModel.lights = lp.getLights();
	}

	MtlParser mp;
	mp.load( swapExtension( file, ".mtl" ) );
	mModel.materials = mp.getMaterials();

	std::vector<std::string> names;

	for( size_t i = 0; i < mModel.materials.size(); i++ ) {
		names.push_back( mModel.materials[i].mtlName );
	}

	int32_t currentMtl = -1;
	std::string raw;

	while( in.good() ) {
		std::getline( in, raw );
		const std::string line = trimmed( raw );
		const char c0 = line.empty() ? '\0' : line[0];
		const char c1 = ( line.size() > 1 ) ? line[1] : '\0';
		const char c2 = ( line.size() > 2 ) ? line[2] : '\0';

		if( c0 == '#' ) {
			continue;
		}

		if( c0 == 'o' ) {
			const std::vector<std::string> parts = tokens( line, " \t" );
			object3D o;
			o.oName = ( parts.size() > 1 ) ? parts[1] : std::string();
			mModel.objects.push_back( o );
		}
		else if( c0 == 'v' ) {
			std::vector<float>* dst = NULL;
			bool isTexture = false;

			if( c1 == ' ' ) {
				dst = &mModel.vertices;
			}
			else if( c1 == 'n' && c2 == ' ' ) {
				dst = &mModel.normals;
			}
			else if( c1 == 't' && c2 == ' ' ) {
				dst = &mModel.textures;
				isTexture = true;
			}

			if( dst != NULL ) {
				const std::vector<std::string> parts = tokens( line, " \t" );
				const size_t n = parts.size();
				dst->push_back( ( n > 1 ) ? toFloat( parts[1] ) : 0.0f );
				dst->push_back( ( n > 2 ) ? toFloat( parts[2] ) : 0.0f );
				dst->push_back( ( n > 3 ) ? toFloat( parts[3] ) : 0.0f );
				(void) isTexture;
			}
		}
		else if( c0 == 'f' ) {
			if( c1 == ' ' ) {
				std::vector<uint32_t> fV, fVN, fVT;
				this->parseFace( line, &fV, &fVN, &fVT );

				mModel.facesV.insert( mModel.facesV.end(), fV.begin(), fV.end() );
				mModel.facesVN.insert( mModel.facesVN.end(), fVN.begin(), fVN.end() );
				mModel.facesVT.insert( mModel.facesVT.end(), fVT.begin(), fVT.end() );
				mModel.facesMtl.push_back( currentMtl );

				if( !mModel.objects.empty() ) {
					object3D& o = mModel.objects.back();
					o.facesV.insert( o.facesV.end(), fV.begin(), fV.end() );
					o.facesVN.insert( o.facesVN.end(), fVN.begin(), fVN.end() );
				}
			}
		}
		else if( line.find( "usemtl" ) != std::string::npos ) {
			const std::vector<std::string> parts = tokens( line, " \t" );
			currentMtl = -1;

			if( parts.size() > 1 ) {
				const std::vector<std::string>::iterator it = std::find( names.begin(), names.end(), parts[1] );
				currentMtl = ( it != names.end() ) ? (int32_t) ( it - names.begin() ) : -1;
			}
		}
	}
}


// source/ModelLoader.cpp:27-41
void ModelLoader::getFacesOfObject( const object3D& object, std::vector<uint4_t>* faces, int32_t offset ) {
	for( size_t i = 0; i + 2 < object.facesV.size(); i += 3 ) {
		const uint4_t f = { object.facesV[i], object.facesV[i + 1], object.facesV[i + 2], (uint32_t) ( offset + (int32_t) faces->size() ) };
		faces->push_back( f );
	}
}


// source/ModelLoader.cpp:44-57
void ModelLoader::getFaceNormalsOfObject( const object3D& object, std::vector<uint4_t>* faceNormals, int32_t offset ) {
	for( size_t i = 0; i + 2 < object.facesVN.size(); i += 3 ) {
		const uint4_t f = { object.facesVN[i], object.facesVN[i + 1], object.facesVN[i + 2], (uint32_t) ( offset + (int32_t) faceNormals->size() ) };
		faceNormals->push_back( f );
	}
}


// source/ModelLoader.cpp:71-88
void ModelLoader::loadModel( std::string filepath, std::string filename ) {
	mObjParser.load( filepath, filename );
}

}  // namespace pbr
