// Host-side scene model: what the reference's ObjParser / MtlParser / LightParser hold after
// ModelLoader::loadModel (source/ObjParser.h:25-29, source/MtlParser.h:42-63,
// source/LightParser.h:20-26).  Plain containers; every consumer (BVH builder, PathTracer
// buffer packing, scene generators) works on this.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace pbr {

struct float4_t { float x, y, z, w; };
struct uint4_t { uint32_t x, y, z, w; };

// object3D, source/ObjParser.h:25-29
struct object3D {
	std::string oName;
	std::vector<uint32_t> facesV;
	std::vector<uint32_t> facesVN;
};

// material_t, source/MtlParser.h:42-63 (defaults: MtlParser.cpp:11-35)
struct material_t {
	std::string mtlName;
	float4_t Ka{ 1.0f, 1.0f, 1.0f, 0.0f };
	float4_t Kd{ 1.0f, 1.0f, 1.0f, 0.0f };
	float4_t Ks{ 1.0f, 1.0f, 1.0f, 0.0f };
	float d = 1.0f;
	float Ni = 1.0f;
	float Ns = 100.0f;
	int8_t illum = 2;
	int8_t light = 0;
	float rough = 1.0f;
	float p = 1.0f;
	float nu = 0.0f;
	float nv = 0.0f;
	float Rs = 0.0f;
	float Rd = 1.0f;
};

// light_t, source/LightParser.h:20-26 (defaults: LightParser.cpp:11-22)
struct light_t {
	std::string lightName;
	uint32_t type = 0;
	float4_t pos{ 1.0f, 1.0f, 1.0f, 0.0f };
	float4_t rgb{ 1.0f, 1.0f, 1.0f, 0.0f };
	float radius = 0.0f;
};

// Everything ObjParser exposes through its getters (source/ObjParser.h:35-44).
struct SceneModel {
	std::vector<float> vertices;     // xyz, flat
	std::vector<float> normals;      // xyz, flat
	std::vector<float> textures;     // uvw, flat
	std::vector<uint32_t> facesV;    // 3 per face
	std::vector<uint32_t> facesVN;   // 3 per face
	std::vector<uint32_t> facesVT;   // 3 per face (when present)
	std::vector<int32_t> facesMtl;   // 1 per face, -1 = none
	std::vector<object3D> objects;
	std::vector<material_t> materials;
	std::vector<light_t> lights;
};

}  // namespace pbr
