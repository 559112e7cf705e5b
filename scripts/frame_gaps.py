"""Where does a single-frame call (pbr_render_frame + pbr_accumulate) spend its time?  Run under
`rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 scripts/frame_gaps.py`, then
`python3 scripts/frame_gaps.py --report <dir>`: per scene, the mean duration of the path-tracing kernel and of foldFrames and
the mean gap between the end of one call's last kernel and the start of the next call's first (host + launch overhead)."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PBR_LAB_ENV", "1")   # lab script: PBR_* variables are mapped onto the library's knobs (package __init__)

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    kinds = [("pathTracing", "path tracing"), ("foldFrames", "foldFrames")]
    seq = [(("trace" if "pathTracing" in r["Kernel_Name"] else "fold" if "foldFrames" in r["Kernel_Name"] else "other"),
            int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    seq = [s for s in seq if s[0] != "other"]
    # the last 400 trace launches = 2 scenes x 200 timed frames
    idx = [i for i, s in enumerate(seq) if s[0] == "trace"]
    for label, lo, hi in (("cornell", -400, -200), ("sponza", -200, None)):
        sel = idx[lo:hi]
        tr = [seq[i][2] - seq[i][1] for i in sel]
        fo = [seq[i + 1][2] - seq[i + 1][1] for i in sel if i + 1 < len(seq) and seq[i + 1][0] == "fold"]
        mid = [seq[i + 1][1] - seq[i][2] for i in sel if i + 1 < len(seq) and seq[i + 1][0] == "fold"]
        gap = [seq[sel[k + 1]][1] - seq[sel[k] + 1][2] for k in range(len(sel) - 1) if seq[sel[k] + 1][0] == "fold"]
        period = [seq[sel[k + 1]][1] - seq[sel[k]][1] for k in range(len(sel) - 1)]
        us = lambda v: sum(v) / max(1, len(v)) / 1e3
        print("%-8s trace %.1f us  trace->fold gap %.1f us  fold %.1f us  fold->next trace gap %.1f us  period %.1f us (n = %d)" % (
            label, us(tr), us(mid), us(fo), us(gap), us(period), len(sel)))
    sys.exit(0)

import numpy as np
import pbr_loader
pbr = pbr_loader.load()
W, H = 1920, 1080
for kind, seed, tris, depth in (("cornell", 1, 0, 8), ("sponza", 2, 260000, 3)):
    pbr.cfg_reset(); pbr.cfg_set(**{"render.max_depth": depth})
    sc = pbr.HostScene.generate(kind, seed, tris)
    dev = pbr.Device(0); dev.upload_scene(sc.desc); dev.configure(sc.config(W, H))
    cam, px = sc.camera(), pbr.pixel_dimension(W, H)
    seeds = pbr.frame_seeds(0, 320)
    for k in range(120):                      # tuner + warm-up
        dev.render_frame(float(seeds[k]), k / (k + 1.0), px, cam); dev.accumulate()
    for k in range(120, 320):
        dev.render_frame(float(seeds[k]), k / (k + 1.0), px, cam); dev.accumulate()
    print(kind, dev.last_plan(), "last kernel ms", dev.last_kernel_ms(), flush=True)
    dev.close()
