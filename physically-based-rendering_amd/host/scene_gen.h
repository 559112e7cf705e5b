// Procedural stand-ins for the scenes BASELINE.json names.  The reference ships no Cornell
// .obj (only binary .blend files) and the Stanford Dragon / Crytek Sponza meshes cannot be
// fetched offline, so every benchmark scene is generated here, seeded and bit-reproducible
// (own integer RNG, own sin/cos — nothing depends on libm), in the same in-memory form
// ObjParser produces.
#pragma once

#include <cstdint>
#include <string>

#include "scene_model.h"

namespace pbr {

struct GeneratedScene {
	SceneModel model;
	// Suggested camera in the reference's convention (Camera.cpp:80-107): eye, and `center`
	// as a view direction with target = eye + (cx, -cy, -cz).
	float eye[3];
	float center[3];
};

// kind: "cornell"  (~40 tris: open-top box, two blocks, 3 objects)
//       "dragon"   (single closed knot mesh in an open box; `triangles` ~ 870k)
//       "sponza"   (colonnaded atrium, many objects; `triangles` ~ 260k)
//       "hairball" (thin random-walk strips in a ball, one object; `triangles` ~ 2M)
// `triangles` = approximate triangle budget (0 = the BASELINE size).  Throws on unknown kind.
GeneratedScene generateScene( const std::string& kind, uint32_t seed, uint32_t triangles );

}  // namespace pbr
