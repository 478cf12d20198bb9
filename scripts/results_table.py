#!/usr/bin/env python3
"""Rewrites BASELINE.md §4 (the results sheet SURVEY.md names) from the round's committed measurements, so that no number in it
is typed by hand or older than the files it cites.

    python scripts/results_table.py r06            # reads profiles/r06/{bench_driver_style.json, pmc_summary*.json, parity_report.json}
    python scripts/results_table.py r06 --check    # fails if BASELINE.md is not what those files generate (a CPU test runs this)

Sources, all under profiles/<round>/:
  bench_driver_style.json   ONE line of `python bench.py --steps 20 --warmup 5` (what the driver runs): the C2 headline, pipeline_fps,
                            option_direct, cpu_baseline and the `configs` legs (C2 with the binary's tiling, C3, C3-literal, C5, C4 on one GPU)
  pmc_summary.json          rocprofv3 --pmc passes of the default run (k_wino): achieved HBM GB/s, MFMA busy share
  pmc_summary_direct.json   the same with the direct kernels pinned (k_pair)
  parity_report.json        tests/test_full_frame_parity.py on the GPU: max LSB and share of differing samples per config and evaluation
The table lands between the markers `<!-- results:begin -->` / `<!-- results:end -->` of BASELINE.md.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_line(path):
    return json.loads([l for l in open(path).read().splitlines() if l.strip().startswith("{")][-1])


def main():
    rnd = next((a for a in sys.argv[1:] if not a.startswith("--")), "r06")
    P = os.path.join(ROOT, "profiles", rnd)
    rel = f"profiles/{rnd}"
    d = load_line(os.path.join(P, "bench_driver_style.json"))
    pmc = json.load(open(os.path.join(P, "pmc_summary.json")))
    pmc_d = json.load(open(os.path.join(P, "pmc_summary_direct.json"))) if os.path.exists(os.path.join(P, "pmc_summary_direct.json")) else {}
    par = json.load(open(os.path.join(P, "parity_report.json")))["cases"] if os.path.exists(os.path.join(P, "parity_report.json")) else {}

    def kern(summary, key):
        ks = [v for k, v in summary.items() if key in k]
        return ks[0] if ks else None

    def pmc_cell(v):
        if not v:
            return "—"
        return (f"{v['_hbm_gbps']:,.0f} GB/s = {v['_hbm_frac_of_8000']:.2f} of 8,000; {(v['FETCH_SIZE'] * 2048 + v['WRITE_SIZE'] * 1024) / 1e6:.1f} MB per launch; "
                f"MFMA pipes {100 * v['_mfma_util']:.1f} % busy, {v['_mfma_tflops_executed']:,.0f} TFLOP/s executed")

    def lsb(prefix):
        out = []
        for ev in ("direct", "winograd"):
            cs = [c for n, c in par.items() if n.startswith(prefix) and n.endswith("_" + ev)]
            if cs:
                out.append(f"{ev}: max {max(c['max_lsb'] for c in cs)} LSB, {100 * max(c['differing_fraction'] for c in cs):.2f} % differ")
        return "; ".join(out) if out else "—"

    rf = d["roofline"]
    cb = d.get("cpu_baseline") or {}
    rows = []
    rows.append(("C1 (256×256 ×2; the GPU path; `tests/test_gpu_parity.py::test_c1_256x256_x2`; plumbing without a GPU: `tests/test_c1_plumbing.py`)", "1", "—", "—", "—", "—", "—", "≤ 1 LSB (test)", "—"))
    rows.append((f"**C2** 1920×1080 → 3840×2160 ×2, whole frame — the headline (`value`)", "1",
                 f"**{d['value']:.1f}**", f"{d['pipeline_fps']:.1f}", f"{d['roofline_frac_whole_path']:.3f}",
                 f"{rf['frac']:.3f} ({rf['launch_us']:.1f} µs, {rf['evaluation'].split(' (')[0]}; κ {rf['kappa']})",
                 pmc_cell(kern(pmc, "k_wino")), lsb("C2_1080p_x2_tile0"),
                 f"{cb.get('value', '—')} ({cb.get('cores', '—')} threads, kind `{cb.get('kind', '—')}`)"))
    od = d.get("option_direct")
    if od:
        rows.append(("C2 with the direct kernels pinned (`REVE_WINOGRAD=0`; the line's `option_direct`)", "1", f"{od['value']:.1f}", "—",
                     f"{od['roofline_frac_whole_path']:.3f}", "—", pmc_cell(kern(pmc_d, "k_pair")), "(as above, `direct`)", ""))
    names = {"C2_tile200": ("C2 with the binary's 200-px tiling (what an unmodified reve gets)", "C2_1080p_x2_tile200"),
             "C3": ("**C3** 1920×1080 → 7680×4320 ×4", "C3_1080p_x4_tile0"), "C3_literal": ("C3 literal \"→4K\": 960×540 → 3840×2160 ×4 (four frames per launch)", "C3literal"),
             "C5": ("**C5** 3840×2160 → 7680×4320 ×2", "C5_4k"), "C4_1gpu": ("**C4** on ONE GPU: the 1080p ×2 stream in 1000-frame segments, each completed before the next", "C4_")}
    for key, (label, pfx) in names.items():
        c = d["configs"][key]
        r = c["roofline"]
        note = f"; slowest stage {c['slowest_stage']}, PCIe bound {c['pcie_bound_fps']:.0f}" if c.get("pcie_bound_fps") else ""
        rows.append((label, "1", f"**{c['value']:.1f}**", f"{c['pipeline_fps']:.1f}{note}", f"{c['roofline_frac_whole_path']:.3f}",
                     f"{r['frac']:.3f} ({r['launch_us']:.1f} µs, {r['evaluation'].split(' (')[0]})", "—", lsb(pfx), ""))
    rows.append(("C4 on 2 / 4 / 8 GPUs", "2 / 4 / 8", "**not measured**: no multi-GPU box was available to the builder; the driver's SCALE run is the measurement. "
                 f"Rehearsed on one GPU over gloo with the driver's command form: `{rel}/bench_N2_dryrun_1gpu_gloo.json`, `bench_N8_dryrun_1gpu_gloo.json`; "
                 f"one rank over real RCCL: `{rel}/bench_one_rank_rccl.json`", "", "", "", "", "16 whole frames per evaluation through a group and through the ring: " + lsb("C4_"), ""))
    head = ("| config | GPUs | frames/s, frames resident in HBM (`value`) | `pipeline_fps` (pinned host → H2D → chain → D2H) | whole-path fraction of the 2.5 PF conv roofline | dominant kernel's `roofline.frac` (launch, evaluation) | "
            "PMC: achieved HBM GB/s, MFMA busy | parity vs the oracle, every sample of whole frames | CPU baseline frames/s |\n|---|---|---|---|---|---|---|---|---|\n")
    body = "".join("| " + " | ".join(str(x) for x in r) + " |\n" for r in rows)
    text = (f"Generated by `scripts/results_table.py {rnd}` from `{rel}/bench_driver_style.json` (one line of `python bench.py --steps 20 --warmup 5`, library "
            f"`{d['library'].get('wino_src_sha256', '')[:12]}…`), `{rel}/pmc_summary.json`, `{rel}/pmc_summary_direct.json` and `{rel}/parity_report.json`.  "
            f"Box-to-box spread on the pool is ±2–4 % (power-capped chip; this round's lines by box: `{rel}/bench_box_spread.txt`, 547.5–568.0 frames/s); the driver's own `BENCH_rNN.json` is the authoritative line.\n\n" + head + body +
            f"\nThe CPU baseline is the oracle (a port, not the reference's CPU path, which does not exist): {cb.get('sample', '—')}.\n")
    path = os.path.join(ROOT, "BASELINE.md")
    s = open(path).read()
    b, e = "<!-- results:begin -->", "<!-- results:end -->"
    if b not in s:
        raise SystemExit("BASELINE.md has no results markers")
    new = s[:s.index(b) + len(b)] + "\n" + text + s[s.index(e):]
    if "--check" in sys.argv:          # tests/test_abi.py: the sheet in the tree IS what the committed measurements generate
        if new != s:
            raise SystemExit(f"BASELINE.md section 4 is not what scripts/results_table.py {rnd} generates from {rel}/: run it")
        print("BASELINE.md section 4 is up to date")
        return
    open(path, "w").write(new)
    print(text)


if __name__ == "__main__":
    main()
