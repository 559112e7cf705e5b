"""Regenerates tests/golden/*.npz from the CPU oracle (python tests/golden/make_golden.py).

These are REGRESSION fixtures: outputs of this repository's own oracle on this repository's own
procedural scenes, frozen so that a toolchain, libm or CPU change that moves a single bit is
noticed — on the build container and on the GPU box alike.  They are NOT reference-derived
pins: the reference ships no golden vectors and its kernels cannot be built here (see
oracle/pt_oracle.h, "PARITY UNPINNED").
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import pbr_loader  # noqa: E402
from oracle import oracle  # noqa: E402

pbr = pbr_loader.load()

CASES = {
    # name: (scene kind, seed, triangles, cfg overrides, width, height, frames)
    "cornell_sa": ("cornell", 1, 0, {"render.max_depth": 4}, 32, 32, 2),
    "cornell_schlick": ("cornell", 1, 0, {"render.max_depth": 4, "render.brdf": 0}, 32, 32, 2),
    "cornell_2spp": ("cornell", 1, 0, {"render.samples": 2}, 32, 24, 2),
    "sponza_small": ("sponza", 2, 3000, {}, 32, 24, 1),
    "hairball_small": ("hairball", 3, 2000, {}, 24, 24, 1),
}


def render_case(name):
    kind, seed, tris, overrides, w, h, frames = CASES[name]
    pbr.cfg_reset()
    pbr.cfg_set(**overrides)
    sc = pbr.HostScene.generate(kind, seed, tris)
    cfg, cam, px = sc.config(w, h), sc.camera(), pbr.pixel_dimension(w, h)
    r = oracle.Renderer(sc.desc, cfg, threads=1)
    img = r.render(0, pbr.frame_seeds(0, frames), px, cam)
    rng = np.random.default_rng(11)
    arr = sc.arrays()
    lo, hi = arr["vertices"][:, :3].min(0) - 0.1, arr["vertices"][:, :3].max(0) + 0.1
    rays = np.concatenate([rng.uniform(lo, hi, (64, 3)), rng.normal(size=(64, 3))], axis=1).astype(np.float32)
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    t, face, normal, counts = oracle.trace_rays(sc.desc, cfg, rays)
    c = r.counter_dict()
    pbr.cfg_reset()
    return {
        "image": img, "debug": r.debug, "counters": np.array([c["nodes"], c["tris"], c["hits"], c["paths"]], np.uint64),
        "bvh": arr["bvh"], "rays": rays, "ray_t": t, "ray_face": face, "ray_counts": counts,
        "px_dim": np.float32(px),
    }


if __name__ == "__main__":
    for name in CASES:
        data = render_case(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
        print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in data.items()})
