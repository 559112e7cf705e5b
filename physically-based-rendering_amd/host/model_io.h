// Scene I/O stand-ins for the reference's ObjParser / MtlParser / LightParser / ModelLoader
// (source/ObjParser.h:32-44, source/MtlParser.h:66-79, source/LightParser.h:29-43,
// source/ModelLoader.h).  Same class and method names, same parsing quirks; the
// implementation is this project's own (no boost).
#pragma once

#include <string>
#include <vector>

#include "scene_model.h"

namespace pbr {

class MtlParser {

	public:
		std::vector<material_t> getMaterials() { return mMaterials; }
		void load( std::string file );

	private:
		std::vector<material_t> mMaterials;

};


class LightParser {

	public:
		std::vector<light_t> getLights() { return mLights; }
		void load( std::string file );

	private:
		std::vector<light_t> mLights;

};


class ObjParser {

	public:
		void load( std::string filepath, std::string filename );
		// Adopt an in-memory scene (procedural generators) instead of parsing a file.
		void adopt( const SceneModel& model ) { mModel = model; }

		std::vector<int32_t> getFacesMtl() { return mModel.facesMtl; }
		std::vector<uint32_t> getFacesV() { return mModel.facesV; }
		std::vector<uint32_t> getFacesVN() { return mModel.facesVN; }
		std::vector<uint32_t> getFacesVT() { return mModel.facesVT; }
		std::vector<light_t> getLights() { return mModel.lights; }
		std::vector<material_t> getMaterials() { return mModel.materials; }
		std::vector<float> getNormals() { return mModel.normals; }
		std::vector<object3D> getObjects() { return mModel.objects; }
		std::vector<float> getTextureCoordinates() { return mModel.textures; }
		std::vector<float> getVertices() { return mModel.vertices; }
		const SceneModel& model() const { return mModel; }

	private:
		void parseFace( const std::string& line, std::vector<uint32_t>* fV, std::vector<uint32_t>* fVN, std::vector<uint32_t>* fVT );

		SceneModel mModel;

};


class ModelLoader {

	public:
		// {a, b, c, global face index}, source/ModelLoader.cpp:27-41
		static void getFacesOfObject( const object3D& object, std::vector<uint4_t>* faces, int32_t offset );
		static void getFaceNormalsOfObject( const object3D& object, std::vector<uint4_t>* faceNormals, int32_t offset );

		ObjParser* getObjParser() { return &mObjParser; }
		void loadModel( std::string filepath, std::string filename );

	private:
		ObjParser mObjParser;

};

}  // namespace pbr
