"""Regenerates tests/golden/ref_*.npz from the REFERENCE'S OWN scene files (container only: needs /root/reference).

    python tests/golden/make_reference_scenes.py

SURVEY.md section 8(c) asks for fixtures of the scenes the reference ships —
resources/models/testing/{pillars,suzanne,spheres}.obj|.mtl|.lights: reference-authored material sets (glass d = 0,
the nu = nv = 100000 lobes of suzanne.mtl / spheres.mtl, `light` flags), a real .lights file with shadow rays — as
flat arrays plus expected outputs; and, beyond what was asked, the other four models the reference ships
(applejack2, applejack3, squirrel-mirror, squirrels).  The .obj / .mtl / .lights files are loaded through this repository's stand-ins for
the reference's loaders (host/model_io.cpp) and its BVH builder replica (host/bvh_builder.cpp); what is stored is DATA:

  inputs    the seven wire-format arrays PathTracer::initOpenCLBuffers would upload (bvh, facesV, facesN, vertices,
            normals, materials, lights), the kernel constants (pbr_config), the camera (camera_cl bytes), pxDim, seeds
  outputs   the oracle's accumulated image, debug image and counters after `frames` frames, and a 4096-ray
            closest-hit batch (t, face, node / face-test counts)

The outputs come from oracle/pt_oracle.c (the reference's kernels cannot run in this image, oracle/pt_oracle.h); the
inputs are the reference's own.  The GPU parity tests render these fixtures on the HIP path (tests/test_gpu_parity.py),
the CPU suite checks that the oracle still reproduces them and — where /root/reference exists — that loader + builder
still produce the stored arrays from the reference's files.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

REFERENCE_MODELS = "/root/reference/resources/models/testing/"

# name: (obj file, Cfg overrides, width, height, frames)
CASES = {
    # glass pillars (d = 0, Ni = 1.5) between coloured walls: refraction at every bounce, added depth
    "ref_pillars_sa": ("pillars.obj", {"render.brdf": 1, "render.max_depth": 4}, 64, 48, 3),
    "ref_pillars_schlick": ("pillars.obj", {"render.brdf": 0, "render.max_depth": 4}, 64, 48, 3),
    # suzanne: 1082 faces, nu = nv = 100000 lobes (pow( x, 1e5 )), 13 materials
    "ref_suzanne_sa": ("suzanne.obj", {"render.brdf": 1}, 64, 48, 3),
    "ref_suzanne_schlick": ("suzanne.obj", {"render.brdf": 0}, 64, 48, 3),
    # ... with suzanne.lights (one orb light) and shadow rays
    "ref_suzanne_sa_shadow": ("suzanne.obj", {"render.brdf": 1, "render.shadow_rays": 1}, 64, 48, 3),
    "ref_suzanne_schlick_shadow": ("suzanne.obj", {"render.brdf": 0, "render.shadow_rays": 1}, 64, 48, 3),
    # spheres: two glass spheres + a mirror-like one in a Cornell box
    "ref_spheres_sa": ("spheres.obj", {"render.brdf": 1, "render.max_depth": 4}, 64, 48, 3),
    "ref_spheres_schlick": ("spheres.obj", {"render.brdf": 0, "render.max_depth": 4}, 64, 48, 3),
    # the rest of what the reference ships (resources/models/testing/README): applejack2 / applejack3 — 8 k faces,
    # 6 / 2 objects (per-object trees under one root), anisotropic lobes (nu = 2, nv = 5), a `light 0` flag
    "ref_applejack2_sa": ("applejack2.obj", {"render.brdf": 1}, 64, 48, 2),
    "ref_applejack3_schlick": ("applejack3.obj", {"render.brdf": 0}, 64, 48, 2),
    # squirrel-mirror: a nu = nv = 400, Rs = 0.95 mirror wall; squirrels: a glass squirrel (d = 0, Ni = 1.5) in a Cornell box
    "ref_squirrel_mirror_sa": ("squirrel-mirror.obj", {"render.brdf": 1, "render.max_depth": 4}, 64, 48, 3),
    "ref_squirrel_mirror_schlick": ("squirrel-mirror.obj", {"render.brdf": 0, "render.max_depth": 4}, 64, 48, 3),
    "ref_squirrels_sa": ("squirrels.obj", {"render.brdf": 1, "render.max_depth": 4}, 64, 48, 3),
    "ref_squirrels_schlick": ("squirrels.obj", {"render.brdf": 0, "render.max_depth": 4}, 64, 48, 3),
}

CONFIG_FIELDS = ("width", "height", "brdf", "shadow_rays", "max_depth", "max_added_depth", "samples", "anti_aliasing", "phong_tessellation")


def load_case(pbr, name):
    obj, overrides, w, h, frames = CASES[name]
    pbr.cfg_reset()
    pbr.cfg_set(**overrides)
    sc = pbr.HostScene.load_obj(REFERENCE_MODELS, obj)
    return sc, sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h), frames


def inputs_of(pbr, sc, cfg, cam, px, frames):
    arr = sc.arrays()
    data = {k: arr[k] for k in ("bvh", "facesV", "facesN", "vertices", "normals", "materials")}
    data["lights"] = arr["lights"][: sc.desc.num_lights]
    data["config"] = np.array([float(getattr(cfg, f)) for f in CONFIG_FIELDS], np.float64)
    data["sky_light"] = np.array(list(cfg.sky_light), np.float32)
    data["camera"] = np.frombuffer(bytes(cam), np.uint8).copy()
    data["px_dim"] = np.float32(px)
    data["seeds"] = pbr.frame_seeds(0, frames)
    return data


def scene_from_fixture(pbr, data):
    """(pbr_scene_desc, pbr_config, pbr_camera, keep-alive) from a fixture's arrays — no loader, no builder, no files."""
    keep = {k: np.ascontiguousarray(data[k]) for k in ("bvh", "facesV", "facesN", "vertices", "normals", "materials", "lights")}
    d = pbr.SceneDesc()
    d.bvh, d.num_nodes = keep["bvh"].ctypes.data, keep["bvh"].shape[0]
    d.facesV, d.facesN, d.num_faces = keep["facesV"].ctypes.data, keep["facesN"].ctypes.data, keep["facesV"].shape[0]
    d.vertices, d.num_vertices = keep["vertices"].ctypes.data, keep["vertices"].shape[0]
    d.normals, d.num_normals = keep["normals"].ctypes.data, keep["normals"].shape[0]
    d.materials, d.num_materials = keep["materials"].ctypes.data, keep["materials"].shape[0]
    d.brdf = 0 if keep["materials"].shape[1] == 12 else 1
    if keep["lights"].shape[0] > 0:
        d.lights, d.num_lights = keep["lights"].ctypes.data, keep["lights"].shape[0]
    else:
        d.lights, d.num_lights = None, 0
    cfg = pbr.Config()
    for f, v in zip(CONFIG_FIELDS, data["config"]):
        setattr(cfg, f, type(getattr(cfg, f))(v))
    for k in range(4):
        cfg.sky_light[k] = float(data["sky_light"][k])
    cfg.tile_world, cfg.tile_rank = 1, 0
    cam = pbr.Camera.from_buffer_copy(np.ascontiguousarray(data["camera"]).tobytes())
    return d, cfg, cam, keep


def ray_batch(data, n=4096, seed=17):
    rng = np.random.default_rng(seed)
    v = data["vertices"][:, :3]
    lo, hi = v.min(0) - 0.2, v.max(0) + 0.2
    rays = np.concatenate([rng.uniform(lo, hi, (n, 3)), rng.normal(size=(n, 3))], axis=1).astype(np.float32)
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    return rays


def outputs_of(pbr, oracle, data):
    desc, cfg, cam, keep = scene_from_fixture(pbr, data)
    r = oracle.Renderer(desc, cfg, threads=os.cpu_count() or 1)
    img = r.render(0, data["seeds"], float(data["px_dim"]), cam)
    c = r.counter_dict()
    rays = ray_batch(data)
    t, face, normal, counts = oracle.trace_rays(desc, cfg, rays)
    return {
        "image": img, "debug": r.debug, "counters": np.array([c["nodes"], c["tris"], c["hits"], c["paths"]], np.uint64),
        "rays": rays, "ray_t": t, "ray_face": face, "ray_counts": counts,
    }


if __name__ == "__main__":
    import pbr_loader
    from oracle import oracle

    pbr = pbr_loader.load()
    for name in CASES:
        sc, cfg, cam, px, frames = load_case(pbr, name)
        data = inputs_of(pbr, sc, cfg, cam, px, frames)
        data.update(outputs_of(pbr, oracle, data))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
        img = data["image"]
        print("%-28s faces %5d nodes %5d mats %2d lights %d  mean rgb %s  counters %s" % (
            name, data["facesV"].shape[0], data["bvh"].shape[0], data["materials"].shape[0], data["lights"].shape[0],
            np.round(img[..., :3].mean((0, 1)), 4), data["counters"].tolist()))
    pbr.cfg_reset()
