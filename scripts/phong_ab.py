"""Phong tessellation (K19) at other register budgets: times the Phong-tessellation build of the lock-step kernel on a UV sphere
with vertex normals over a floor.   PBR_LAB_ENV=1 PBR_HIP_LIB=lab/libpbrhip_ph4.so python3 scripts/phong_ab.py"""
import hashlib, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)
import numpy as np
import pbr_loader
pbr = pbr_loader.load()


def smooth_scene(path, rings, segs):
    verts, norms, faces = [], [], []
    for i in range(rings + 1):
        th = np.pi * i / rings
        for j in range(segs):
            ph = 2 * np.pi * j / segs
            n = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
            verts.append(0.6 * n + [0.0, 0.75, 0.0]); norms.append(n)
    for i in range(rings):
        for j in range(segs):
            a, b = i * segs + j, i * segs + (j + 1) % segs
            c, d = a + segs, b + segs
            if i > 0:
                faces.append((a, b, c))
            if i < rings - 1:
                faces.append((b, d, c))
    lines = ["mtllib s.mtl", "o Ball"]
    lines += ["v %.9g %.9g %.9g" % tuple(v) for v in verts]
    lines += ["vn %.9g %.9g %.9g" % tuple(n) for n in norms]
    lines += ["usemtl Shiny"] + ["f %d//%d %d//%d %d//%d" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1) for a, b, c in faces]
    k, kn = len(verts), len(norms)
    lines += ["o Floor", "v -2 0 -2", "v 2 0 -2", "v 2 0 2", "v -2 0 2", "vn 0 1 0", "usemtl Matte",
              "f %d//%d %d//%d %d//%d" % (k + 1, kn + 1, k + 3, kn + 1, k + 2, kn + 1),
              "f %d//%d %d//%d %d//%d" % (k + 1, kn + 1, k + 4, kn + 1, k + 3, kn + 1)]
    open(os.path.join(path, "s.obj"), "w").write("\n".join(lines) + "\n")
    open(os.path.join(path, "s.mtl"), "w").write("newmtl Shiny\nKd 0.8 0.3 0.2\nKs 0.9 0.9 0.9\nnu 200\nnv 200\nRs 0.4\nRd 0.6\n\nnewmtl Matte\nKd 0.6 0.6 0.7\n\nnewmtl sky_light\nKd 0.9 0.95 1.0\n")
    return pbr.HostScene.load_obj(path + "/", "s.obj")


W, H = 1920, 1080
tag = os.path.basename(os.environ.get("PBR_HIP_LIB", "default"))
for rings, segs in ((10, 16), (48, 96)):
    with tempfile.TemporaryDirectory() as tmp:
        pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": 4, "render.brdf": 1, "render.phong_tessellation": 0.6})
        sc = smooth_scene(tmp, rings, segs)
    cfg, cam, px = sc.config(W, H), sc.camera(), pbr.pixel_dimension(W, H)
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(cfg)
    dev.render(0, pbr.frame_seeds(0, 8), px, cam)
    best = 1e9
    for rep in range(3):
        dev.reset_accum(); dev.render(0, pbr.frame_seeds(0, 32), px, cam); best = min(best, dev.last_kernel_ms())
    digest = hashlib.sha1(np.ascontiguousarray(dev.read_output()).tobytes()).hexdigest()[:12]
    print("%-22s sphere %3d x %3d  %-20s 32 frames %8.2f ms  %7.1f Msamples/s  sha1 %s" % (tag, rings, segs, dev.last_plan()[0], best, W * H * 32 / best / 1e3, digest), flush=True)
    dev.close()
