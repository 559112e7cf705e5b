"""Scene I/O stand-ins and buffer packing: the parsing quirks of source/ObjParser.cpp,
MtlParser.cpp, LightParser.cpp and the wire formats of PathTracer::initOpenCLBuffers_*.
CPU only; every input is written by the test (plus the reference's own assets when present)."""
import os

import numpy as np
import pytest

from conftest import REFERENCE_MODELS

OBJ = """# a comment
mtllib whatever.mtl
o First
v 0 0 0
v 1 0 0
v 0 1 0
v 0 0 1
vn 0 0 1
vt 0.5 0.5
usemtl Red
f 1//1 2//1 3//1
f 1/1/1 3/1/1 4/1/1
usemtl Missing
f 1 2 4
o Second
usemtl Glass
f 2/1 3/1 4/1
"""

MTL = """# materials
newmtl Red
Kd 0.8 0.1 0.1
Ni 1.0
d 1.0
Tr 0.75
nu 10
nv 20
Rs 0.25
Rd 0.5
rough 0.3
p 0.9

newmtl Glass
Tr 0.4
d 0.1
Ni 1.5

newmtl sky_light
Kd 0.1234567 0.5 0.9999999
"""

LIGHTS = """newlight orb0
type 2
pos 0.1 1.3 1.2
rgb 0.846 0.933 0.949
radius 0.1
newlight p1
type 1
pos 1 2 3
"""


@pytest.fixture()
def model_dir(tmp_path):
    (tmp_path / "m.obj").write_text(OBJ)
    (tmp_path / "m.mtl").write_text(MTL)
    (tmp_path / "m.lights").write_text(LIGHTS)
    return str(tmp_path) + "/"


def test_obj_quirks_and_packing(cfg_defaults, model_dir):
    pbr = cfg_defaults
    sc = pbr.HostScene.load_obj(model_dir, "m.obj")
    arr, info = sc.arrays(), sc.info
    assert info["objects"] == 2 and info["faces"] == 4 and info["vertices"] == 4 and info["materials"] == 3
    assert info["lights"] == 0                                   # shadow_rays = 0: .lights is not even read

    # faces travel in leaf order with the material id in .w; the face after `usemtl Missing` has -1
    mats = sorted(arr["facesV"][:, 3].tolist())
    assert mats == [0, 0, 1, 0xFFFFFFFF]
    tri = {tuple(sorted(f[:3].tolist())) for f in arr["facesV"]}
    assert tri == {(0, 1, 2), (0, 2, 3), (0, 1, 3), (1, 2, 3)}      # 1-based -> 0-based; "v/vt" read as v//vn keeps v

    # vec3 -> float4 with w = 0
    assert np.array_equal(arr["vertices"], np.array([[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float32))

    # Shirley-Ashikhmin layout {d, Ni, nu, nv, Rs, Rd, -, -}, Kd, Ks; `Tr` ignored once any `d` was seen
    m = arr["materials"]
    assert m.shape == (3, 16)
    assert np.allclose(m[0, :6], [1.0, 1.0, 10, 20, 0.25, 0.5]) and np.allclose(m[0, 8:11], [0.8, 0.1, 0.1])
    assert np.allclose(m[1, :2], [0.1, 1.5])                      # Glass: Tr came BEFORE d in its block but after Red's d -> ignored
    assert np.allclose(m[2, 8:11], [0.1234567, 0.5, 0.9999999])

    # SKY_LIGHT goes through "%f": rounded to 6 decimals (PathTracer.cpp:470-472)
    cfg = sc.config(64, 64)
    assert list(cfg.sky_light)[:3] == [np.float32(0.123457), np.float32(0.5), np.float32(1.0)]
    assert (cfg.brdf, cfg.max_depth, cfg.max_added_depth, cfg.samples, cfg.shadow_rays) == (1, 3, 5, 1, 0)
    assert cfg.anti_aliasing == np.float32(0.7)


def test_schlick_material_layout(cfg_defaults, model_dir):
    pbr = cfg_defaults
    pbr.cfg_set(**{"render.brdf": 0})
    m = pbr.HostScene.load_obj(model_dir, "m.obj").arrays()["materials"]
    assert m.shape == (3, 12)
    assert np.allclose(m[0, :4], [1.0, 1.0, 0.9, 0.3])            # d, Ni, p, rough
    assert np.allclose(m[0, 4:7], [0.8, 0.1, 0.1]) and np.allclose(m[0, 8:11], [1, 1, 1])


def test_lights_only_with_shadow_rays(cfg_defaults, model_dir):
    pbr = cfg_defaults
    pbr.cfg_set(**{"render.shadow_rays": 1})
    sc = pbr.HostScene.load_obj(model_dir, "m.obj")
    assert sc.info["lights"] == 2
    lights = sc.arrays()["lights"]
    assert np.allclose(lights[0], [0.1, 1.3, 1.2, 0, 0.846, 0.933, 0.949, 0, 2, 0.1, 0, 0])
    assert np.allclose(lights[1], [1, 2, 3, 0, 1, 1, 1, 0, 1, 0, 0, 0])
    assert sc.config(8, 8).shadow_rays == 1


def test_missing_lights_file_switches_shadow_rays_off(cfg_defaults, model_dir):
    """LightParser.cpp:119-121: a .lights file without lights resets render.shadow_rays to 0."""
    pbr = cfg_defaults
    pbr.cfg_set(**{"render.shadow_rays": 1})
    open(model_dir + "m.lights", "w").write("# nothing here\n")
    sc = pbr.HostScene.load_obj(model_dir, "m.obj")
    assert sc.info["lights"] == 0
    assert pbr.cfg_get("render.shadow_rays") == "0"
    assert sc.config(8, 8).shadow_rays == 0


def test_missing_file_is_an_error_not_a_crash(cfg_defaults, tmp_path):
    pbr = cfg_defaults
    with pytest.raises(pbr.PbrError):
        pbr.HostScene.load_obj(str(tmp_path), "nope.obj")


def test_cfg_defaults_match_reference_config_json(cfg_defaults):
    pbr = cfg_defaults
    expect = {
        "accel_struct": "0", "bvh.max_faces": "2", "bvh.sah_faces_limit": "100000", "bvh.skip_ahead": "true",
        "bvh.skip_ahead_compare": "0.7", "render.antialiasing": "0.7", "render.brdf": "1",
        "render.max_added_depth": "5", "render.max_depth": "3", "render.phong_tessellation": "0.0",
        "render.samples": "1", "render.shadow_rays": "0", "window.width": "800", "window.height": "600",
        "camera.eye.y": "1.0", "camera.eye.z": "3.0", "camera.perspective.fov": "45.0",
        "camera.thin_lense.aperture": "1.8", "camera.thin_lense.focal_length": "0.035",
    }
    for key, value in expect.items():
        assert pbr.cfg_get(key) == value, key


def test_cfg_reads_json_with_comments(cfg_defaults, tmp_path):
    pbr = cfg_defaults
    path = tmp_path / "config.json"
    path.write_text('{\n // comment\n "render": { "max_depth": 7, // trailing\n "brdf": 0 },\n "bvh": { "skip_ahead": false },\n "import_path": "/a b/" }\n')
    assert pbr.host.pbrh_cfg_load(str(path).encode()) == 0
    assert pbr.cfg_get("render.max_depth") == "7" and pbr.cfg_get("render.brdf") == "0"
    assert pbr.cfg_get("bvh.skip_ahead") == "false" and pbr.cfg_get("import_path") == "/a b/"
    assert pbr.cfg_get("render.samples") == "1"                  # untouched keys keep their defaults


def test_default_camera_and_pixel_size(cfg_defaults):
    """updateEyeBuffer (PathTracer.cpp:628-652) on the default pose (config.json:3-18): eye (0,1,3)
    looking down -Z; initKernelArgs' pixel size (PathTracer.cpp:89-91)."""
    pbr = cfg_defaults
    sc = pbr.HostScene.generate("cornell")
    cam = sc.camera()
    assert (cam.eye.x, cam.eye.y, cam.eye.z) == (0.0, 1.0, 3.0)
    assert (cam.w.x, cam.w.y, cam.w.z) == (0.0, 0.0, -1.0)
    assert (cam.u.x, abs(cam.u.y), cam.u.z) == (1.0, 0.0, 0.0)
    assert (cam.v.x, cam.v.y, cam.v.z) == (0.0, 1.0, 0.0)
    assert tuple(cam.focusPoint) == (-1, -1)
    assert tuple(cam.lense) == (np.float32(0.035), np.float32(1.8))
    px = pbr.pixel_dimension(800, 600, 45.0)
    assert px == pytest.approx((800 / 600) * 2 * np.tan(np.radians(45.0) / 2) / 800, rel=1e-6)


@pytest.mark.skipif(not os.path.isdir(REFERENCE_MODELS), reason="reference assets only exist in the build container")
@pytest.mark.parametrize("name,faces,objects", [
    ("pillars.obj", 56, 5), ("spheres.obj", 800, 4), ("squirrel-mirror.obj", 1020, 3),
    ("squirrels.obj", 1408, 3), ("suzanne.obj", 1082, 10), ("applejack3.obj", 8068, 2),
])
def test_reference_assets_load(cfg_defaults, name, faces, objects):
    pbr = cfg_defaults
    sc = pbr.HostScene.load_obj(REFERENCE_MODELS, name)
    assert sc.info["faces"] == faces and sc.info["objects"] == objects
    arr = sc.arrays()
    assert arr["facesV"][:, :3].max() < sc.info["vertices"]


def test_image_writers_round_trip(cfg_defaults, tmp_path):
    """pbrh_write_ppm / pbrh_write_pfm: the display image and the linear image as files."""
    pbr = cfg_defaults
    rng = np.random.default_rng(2)
    rgba8 = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    pbr.write_ppm(tmp_path / "a.ppm", rgba8)
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6\n7 5\n255\n")
    assert np.array_equal(np.frombuffer(raw[len(b"P6\n7 5\n255\n"):], np.uint8).reshape(5, 7, 3), rgba8[..., :3])
    lin = rng.normal(size=(4, 6, 4)).astype(np.float32)
    pbr.write_pfm(tmp_path / "b.pfm", lin)
    raw = (tmp_path / "b.pfm").read_bytes()
    head = b"PF\n6 4\n-1.0\n"
    assert raw.startswith(head)
    assert np.array_equal(np.frombuffer(raw[len(head):], "<f4").reshape(4, 6, 3), lin[..., :3])
    with pytest.raises(pbr.PbrError):
        pbr.write_ppm(tmp_path / "no" / "dir.ppm", rgba8)


def test_png_and_exr_writers(pbr, tmp_path):
    """pbrh_write_png / pbrh_write_exr — the formats SURVEY.md section 8(f) row 4 names — read back here by a decoder
    written from the two specifications (zlib for the PNG's stream; the EXR is uncompressed): every pixel, the row
    order (PNG: top row first as read_display returns it; EXR: the input's row 0 is the BOTTOM, the file's first scanline
    the top), chunk CRCs, the EXR's header attributes and offset table."""
    import struct
    import zlib
    rng = np.random.default_rng(5)
    w, h = 37, 21
    rgba8 = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    pbr.write_png(tmp_path / "a.png", rgba8)
    raw = (tmp_path / "a.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(raw):
        n, kind = struct.unpack(">I4s", raw[at:at + 8])
        body = raw[at + 8:at + 8 + n]
        assert struct.unpack(">I", raw[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + body)
        chunks.append((kind, body))
        at += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 6, 0, 0, 0)
    lines = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 4 * w)
    assert (lines[:, 0] == 0).all() and np.array_equal(lines[:, 1:].reshape(h, w, 4), rgba8)
    # more than one stored deflate block (65535 bytes each)
    big = rng.integers(0, 256, (200, 160, 4), dtype=np.uint8)
    pbr.write_png(tmp_path / "big.png", big)
    rawb = (tmp_path / "big.png").read_bytes()
    idat = rawb[rawb.index(b"IDAT") + 4:]
    assert np.array_equal(np.frombuffer(zlib.decompress(idat[:-16]), np.uint8).reshape(200, 1 + 640)[:, 1:].reshape(200, 160, 4), big)

    lin = rng.normal(size=(h, w, 4)).astype(np.float32)
    lin[3, 4] = [np.inf, 0.0, -0.0, 1e-40]
    pbr.write_exr(tmp_path / "b.exr", lin)
    raw = (tmp_path / "b.exr").read_bytes()
    assert struct.unpack("<II", raw[:8]) == (20000630, 2)
    at, attrs = 8, {}
    while raw[at] != 0:
        end = raw.index(b"\0", at); name = raw[at:end].decode(); at = end + 1
        end = raw.index(b"\0", at); kind = raw[at:end].decode(); at = end + 1
        n = struct.unpack("<i", raw[at:at + 4])[0]
        attrs[name] = (kind, raw[at + 4:at + 4 + n]); at += 4 + n
    at += 1
    assert set(attrs) == {"channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio", "screenWindowCenter", "screenWindowWidth"}
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    assert struct.unpack("<4i", attrs["dataWindow"][1]) == (0, 0, w - 1, h - 1) and attrs["displayWindow"] == attrs["dataWindow"]
    ch, names = attrs["channels"][1], []
    while ch[0] != 0:
        end = ch.index(b"\0"); names.append(ch[:end].decode())
        assert struct.unpack("<i4xii", ch[end + 1:end + 17]) == (2, 1, 1)       # FLOAT, no subsampling
        ch = ch[end + 17:]
    assert names == ["A", "B", "G", "R"]
    offsets = struct.unpack("<%dQ" % h, raw[at:at + 8 * h])
    got = np.zeros((h, w, 4), np.float32)
    for y, off in enumerate(offsets):
        line_y, size = struct.unpack("<ii", raw[off:off + 8])
        assert (line_y, size) == (y, w * 16)
        planes = np.frombuffer(raw[off + 8:off + 8 + size], np.float32).reshape(4, w)
        got[h - 1 - y] = planes[[3, 2, 1, 0]].T                                  # A, B, G, R -> R, G, B, A; file row 0 = the top
    assert offsets[-1] + 8 + w * 16 == len(raw)
    assert np.array_equal(got.view(np.uint32), lin.view(np.uint32))                 # bit for bit, -0 and the denormal included
    with pytest.raises(pbr.PbrError):
        pbr.write_png(tmp_path / "no" / "dir.png", rgba8)
    with pytest.raises(pbr.PbrError):
        pbr.write_exr(tmp_path / "no" / "dir.exr", lin)


def test_mode_keys_reach_the_configuration(cfg_defaults):
    """"hip.traversal" / "hip.arith" (not reference keys; absent = 0 = the reference's behaviour) are how a viewer that keeps
    the reference's config.json switches the HIP core's two opt-in modes: PathTracer::makeConfig copies them into pbr_config."""
    pbr = cfg_defaults
    sc = pbr.HostScene.generate("cornell", 1, 0)
    cfg = sc.config(64, 64)
    assert (cfg.traversal, cfg.arith) == (0, 0)
    pbr.cfg_set(**{"hip.traversal": 2, "hip.arith": 1})
    cfg = sc.config(64, 64)
    assert (cfg.traversal, cfg.arith) == (2, 1)
    pbr.cfg_reset()
    assert (sc.config(64, 64).traversal, sc.config(64, 64).arith) == (0, 0)
