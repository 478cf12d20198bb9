#!/usr/bin/env python3
"""From "a maintainer has the release zip" to "parity pinned" in one command.

    python scripts/pin_against_binary.py --model-dir MODELS --frames IN_DIR --binary-out OUT_DIR [--scale 2]
    python scripts/pin_against_binary.py --model-dir MODELS --frames IN_DIR --binary /path/to/realesrgan-ncnn-vulkan    (runs it first)

What it needs — none of which exists in /root/reference or in this image (reve's models are git-ignored,
reve-gui/.gitignore:27-30; the binary ships only inside the release zip, README.md:27-29):
  MODELS   the directory with realesr-animevideov3-x{2,3,4}.param/.bin  (what reve's `-n` names, reve-shared/src/lib.rs:140-141)
  IN_DIR   PNG frames (what reve writes to tmp_frames/<i>/frame%08d.png, lib.rs:93)
  OUT_DIR  the PNGs the ORIGINAL realesrgan-ncnn-vulkan wrote for them (`-i IN_DIR -o OUT_DIR -n realesr-animevideov3-x2 -s 2 -f png`,
           the command line of lib.rs:134-147; run wherever a Vulkan GPU is), same file names

What it does:
  1. per frame, the LSB histogram of every evaluation this repository has against the binary's bytes: the HIP path with the direct
     pair kernels and with the Winograd pairs (needs an MI355X; --no-gpu skips them) and the CPU oracle's modes 1-4 (fp16 storage
     with direct sums / Winograd F(2x2,3x3) / F(4x4,3x3) / F(2,3) along the row) — which of them IS the binary's arithmetic is
     the open question of SURVEY.md §8(c), and this table answers it;
  2. the tile size the binary's output is consistent with (the oracle run with --tiles candidates; seams betray the tiling);
  3. the model's conditioning estimate kappa and the evaluation the library's default (auto) chooses for it (reve_model_report);
  4. writes tests/golden/binary_pins/<name>.npz — inputs, the binary's outputs, the tile size found, sha256 of the model files.
     DATA ONLY: never the model, never the binary.  tests/test_oracle.py::test_binary_pins (CPU oracle) and
     tests/test_gpu_parity.py::test_binary_pins (HIP path) consume every pin present when REVE_MODEL_DIR names the model the pin's
     digest matches — from then on the oracle is pinned to reference-held vectors and `parity: partial` can be re-judged.

The report goes to stdout and to <pins dir>/<name>.report.json.  Exit status 0 = some evaluation is within --tolerance LSB of the
binary on every frame; 1 = none is (the numbers say how far); 2 = bad input.
"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ORACLE_MODES = {"oracle_mode1_fp16_direct": 1, "oracle_mode2_fp16_winograd_f2x2": 2, "oracle_mode3_fp16_winograd_f4x4": 3,
                "oracle_mode4_fp16_winograd_row_f23": 4}


def lsb_histogram(out, ref):
    d = np.abs(out.astype(np.int16) - ref.astype(np.int16))
    hist = np.bincount(d.reshape(-1))
    return {"max_lsb": int(d.max()), "differing_fraction": float((d > 0).mean()), "mean_abs": float(d.mean()),
            "histogram": {str(i): int(n) for i, n in enumerate(hist) if n}}


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def read_png(path):
    """8-bit RGB through the library's own decoder (RGBA / 16-bit are reduced the way the single-file path reduces them)"""
    from reve_amd.upscaler import png_read
    return png_read(path)


def model_report(model_dir, name, scale):
    import ctypes as C
    from reve_amd import _lib
    buf = C.create_string_buffer(1 << 16)
    rc = _lib.load().reve_model_report(model_dir.encode(), name.encode(), scale, buf, len(buf))
    if rc != 0:
        raise SystemExit(f"model report failed: {_lib.load().reve_last_error(None).decode()}")
    return json.loads(buf.value.decode())


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model-dir", required=True)
    ap.add_argument("--model-name", default="realesr-animevideov3", help="as reve names it; -x<scale> is resolved like the binary does")
    ap.add_argument("--scale", type=int, default=2, choices=[2, 3, 4])
    ap.add_argument("--frames", required=True, help="directory of input PNGs")
    ap.add_argument("--binary-out", default=None, help="directory of the PNGs the original binary wrote (same names)")
    ap.add_argument("--binary", default=None, help="instead of --binary-out: the original executable itself (where it can run: a Vulkan GPU); it is started "
                                                   "once with reve's own argv (reve-shared/src/lib.rs:134-147) and writes into a temporary directory")
    ap.add_argument("--tiles", default="0,100,200,400", help="tile sizes to test the binary's output against (0 = whole frame); the binary's own choice is 200 on a large GPU, 100 / 32 on small ones")
    ap.add_argument("--tile", type=int, default=None, help="skip the search: the binary was run with -t N")
    ap.add_argument("--max-frames", type=int, default=4, help="frames examined and stored (a 1080p pin is ~10 MB compressed)")
    ap.add_argument("--pins-dir", default=os.path.join(ROOT, "tests", "golden", "binary_pins"))
    ap.add_argument("--name", default=None, help="name of the pin file (default: <model>_<first frame>)")
    ap.add_argument("--no-gpu", action="store_true", help="oracle only (no MI355X on this machine)")
    ap.add_argument("--tolerance", type=int, default=1, help="LSB per RGB sample (north_star's +-1)")
    ap.add_argument("--stand-in", default=None, help="say in the pin that --binary-out was NOT written by the real binary (rehearsals)")
    args = ap.parse_args(argv)

    if (args.binary_out is None) == (args.binary is None):
        print("give exactly one of --binary-out DIR and --binary EXE", file=sys.stderr)
        return 2
    if args.binary:
        import subprocess
        import tempfile
        args.binary_out = tempfile.mkdtemp(prefix="binary_out_")
        # the command line of reve-shared/src/lib.rs:134-147, with the model that matches the scale (reve's own always-x2 name is the
        # bug of SURVEY.md 9.1-A: with it the binary would run the x2 graph on an x3 / x4 canvas)
        cmd = [args.binary, "-i", args.frames, "-o", args.binary_out, "-n", f"realesr-animevideov3-x{args.scale}", "-s", str(args.scale), "-f", "png",
               "-m", args.model_dir] + (["-t", str(args.tile)] if args.tile is not None else [])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"{' '.join(cmd)} failed ({r.returncode}): {r.stderr[-500:]}", file=sys.stderr)
            return 2
    from oracle import ref
    from reve_amd import ncnn_io
    import ctypes as C
    from reve_amd import _lib
    res = C.create_string_buffer(512)
    if _lib.load().reve_resolve_model_name(args.model_name.encode(), args.scale, res, len(res)) < 0:
        raise SystemExit("bad model name")
    name = res.value.decode()
    files = [os.path.join(args.model_dir, name + ext) for ext in (".param", ".bin")]
    for f in files:
        if not os.path.exists(f):
            print(f"missing model file {f}", file=sys.stderr)
            return 2
    param, binb = ncnn_io.read_model_files(args.model_dir, name)
    weights = ncnn_io.parse_model(param.decode(), binb)
    if weights["scale"] != args.scale:
        print(f"{name}: the model's PixelShuffle factor is {weights['scale']}, --scale {args.scale}", file=sys.stderr)
        return 2
    digests = {os.path.basename(f): sha256_file(f) for f in files}
    rep = model_report(args.model_dir, args.model_name, args.scale)

    names = sorted(n for n in os.listdir(args.frames) if n.lower().endswith(".png") and os.path.exists(os.path.join(args.binary_out, n)))
    if not names:
        print("no PNG present in both --frames and --binary-out", file=sys.stderr)
        return 2
    names = names[:args.max_frames]
    frames = [read_png(os.path.join(args.frames, n)) for n in names]
    theirs = [read_png(os.path.join(args.binary_out, n)) for n in names]
    for n, a, b in zip(names, frames, theirs):
        if b.shape != (a.shape[0] * args.scale, a.shape[1] * args.scale, 3):
            print(f"{n}: the binary's output is {b.shape[1]}x{b.shape[0]}, expected {a.shape[1] * args.scale}x{a.shape[0] * args.scale}", file=sys.stderr)
            return 2

    # ---- 2. which tiling?  (oracle mode 1 per candidate; the right one has no seams: the fewest differing samples)
    if args.tile is not None:
        tiles, tile_table = [args.tile], None
    else:
        tiles = sorted({int(t) for t in args.tiles.split(",")})
    tile_scores = {}
    for t in tiles:
        hs = [lsb_histogram(ref.upscale(weights, f, tile=t, prepad=10), o) for f, o in zip(frames[:2], theirs[:2])]
        tile_scores[t] = {"differing_fraction": float(np.mean([h["differing_fraction"] for h in hs])), "max_lsb": max(h["max_lsb"] for h in hs)}
    best_tile = min(tile_scores, key=lambda t: (tile_scores[t]["differing_fraction"], tile_scores[t]["max_lsb"]))
    tile_table = {str(t): v for t, v in tile_scores.items()}

    # ---- 1. every evaluation against the binary, at that tiling
    evals = {}
    for label, mode in ORACLE_MODES.items():
        evals[label] = [lsb_histogram(ref.upscale(weights, f, mode=mode, tile=best_tile, prepad=10), o) for f, o in zip(frames, theirs)]
    gpu_note = None
    if not args.no_gpu:
        try:
            from reve_amd.upscaler import Upscaler
            with Upscaler(args.scale, param=param, bin=binb, tile=best_tile) as up:
                for label, w in (("hip_direct", 0), ("hip_winograd", 1)):
                    up.set_option("winograd", w)
                    evals[label] = [lsb_histogram(up.upscale(f), o) for f, o in zip(frames, theirs)]
                up.set_option("winograd", 2)
                gpu_note = f"auto chose {'winograd' if up.get_option('winograd') else 'direct'} (kappa {up.get_option('winograd_kappa_permille') / 1000:.3f})"
        except Exception as e:   # noqa: BLE001
            gpu_note = f"HIP path not run: {e}"
    else:
        gpu_note = "HIP path not run (--no-gpu)"

    summary = {k: {"max_lsb": max(h["max_lsb"] for h in v), "worst_differing_fraction": max(h["differing_fraction"] for h in v),
                   "within_tolerance": all(h["max_lsb"] <= args.tolerance for h in v)} for k, v in evals.items()}
    closest = min(summary, key=lambda k: (summary[k]["max_lsb"], summary[k]["worst_differing_fraction"]))
    pinned = any(v["within_tolerance"] for v in summary.values())
    report = {"model": name, "scale": args.scale, "model_sha256": digests, "kappa": rep["kappa"], "kappa_limit": rep["kappa_limit"],
              "evaluation_auto_would_choose": rep["evaluation_auto_would_choose"], "frames": names,
              "frame_sizes": [[int(f.shape[1]), int(f.shape[0])] for f in frames],
              "tile_search": tile_table, "tile_consistent_with_the_binary": best_tile, "tolerance_lsb": args.tolerance,
              "summary": summary, "closest_evaluation": closest, "per_frame": evals, "hip": gpu_note,
              "binary_out_is": args.stand_in or "written by the original realesrgan-ncnn-vulkan (the maintainer's statement)",
              "verdict": ("PINNED: " + ", ".join(k for k, v in summary.items() if v["within_tolerance"]) + f" within {args.tolerance} LSB of the binary on every frame")
                         if pinned else f"NOT within {args.tolerance} LSB: closest is {closest} at {summary[closest]['max_lsb']} LSB"}

    # ---- 4. the pin: data only
    os.makedirs(args.pins_dir, exist_ok=True)
    pin_name = args.name or f"{name}_{os.path.splitext(names[0])[0]}"
    meta = {"model": name, "scale": args.scale, "tile": best_tile, "prepad": 10, "model_sha256": digests, "frames": names,
            "binary_out_is": report["binary_out_is"], "kappa": rep["kappa"], "made_by": "scripts/pin_against_binary.py"}
    arrays = {"meta": np.array(json.dumps(meta))}
    for i, (f, o) in enumerate(zip(frames, theirs)):
        arrays[f"img_{i}"], arrays[f"out_{i}"] = f, o
    np.savez_compressed(os.path.join(args.pins_dir, pin_name + ".npz"), **arrays)
    with open(os.path.join(args.pins_dir, pin_name + ".report.json"), "w") as fh:
        json.dump(report, fh, indent=1)

    print(json.dumps({k: report[k] for k in ("model", "kappa", "evaluation_auto_would_choose", "tile_search", "tile_consistent_with_the_binary", "summary",
                                              "closest_evaluation", "hip", "verdict")}, indent=1))
    print(f"pin written: {os.path.join(args.pins_dir, pin_name + '.npz')} ({len(names)} frame(s)); run the suites with REVE_MODEL_DIR={args.model_dir}")
    return 0 if pinned else 1


if __name__ == "__main__":
    sys.exit(main())
