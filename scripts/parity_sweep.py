"""Parity of the HIP path against the oracle over WEIGHT STATISTICS: every draw of
reve_amd.synth.WEIGHT_DRAWS x the x2 / x3 / x4 graphs x an S-toon and an S-noise frame, through the direct pairs (option "winograd" 0),
the Winograd pairs (1) and what the library ships as its default (auto: the rule of DESIGN.md §3 decides from the weights), against oracle
mode 1 (direct sums, fp16 storage).  Prints a table and
writes gpurun_out/parity_sweep.json (copy to profiles/rNN/).  env: W, H (512 x 288), SCALES ("2,3,4")."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reve_amd import synth, ncnn_io
from reve_amd.upscaler import Upscaler
from oracle import ref


def sweep(W=512, H=288, scales=(2, 3, 4), draws=None, log=print):
    rows = []
    for name in (draws or synth.WEIGHT_DRAWS):
        for scale in scales:
            w = synth.make_weights_draw(scale, name)
            p, b = ncnn_io.build_param_text(scale).encode(), ncnn_io.build_bin(w)
            with Upscaler(scale, param=p, bin=b) as up, Upscaler(scale, param=p, bin=b) as wi, Upscaler(scale, param=p, bin=b) as au:
                up.set_option("winograd", 0)
                wi.set_option("winograd", 1)
                assert au.get_option("winograd_mode") == 2          # auto is the default: the library's rule decides from the weights
                kappa, chose = au.get_option("winograd_kappa_permille") / 1000.0, au.get_option("winograd")
                acts = [float(np.abs(up.debug_layer(synth.toon_frame(7, 128, 96), L)).max()) for L in (0, 4, 8, 12, 16)]
                for kind, img in (("toon", synth.toon_frame(7, W, H)), ("noise", synth.noise_frame(7, W, H))):
                    exp = ref.upscale(w, img).astype(np.int32)
                    r = {"draw": name, "scale": scale, "frame": kind, "kappa": kappa, "auto_chose_winograd": bool(chose), "max_abs_activation_L0_4_8_12_16": [round(a, 3) for a in acts],
                         "saturated_fraction": round(float(((exp == 0) | (exp == 255)).mean()), 4)}
                    for path, u in (("direct", up), ("winograd", wi), ("auto", au)):
                        d = np.abs(u.upscale(img).astype(np.int32) - exp)
                        r[path] = {"max_lsb": int(d.max()), "fraction_differing": round(float((d > 0).mean()), 6),
                                   "lsb_histogram": [int(x) for x in np.bincount(d.ravel(), minlength=2)][:8]}
                    rows.append(r)
                    log(f"{name:22s} x{scale} {kind:5s} max|act| {acts[-1]:9.3g} (peak {max(acts):9.3g}) sat {r['saturated_fraction']:.2f} | direct: max {r['direct']['max_lsb']} LSB, "
                        f"{100 * r['direct']['fraction_differing']:.3f} % differ | winograd: max {r['winograd']['max_lsb']} LSB, {100 * r['winograd']['fraction_differing']:.3f} % differ"
                        f" | kappa {kappa:.3f} -> auto {'winograd' if chose else 'direct'}: max {r['auto']['max_lsb']} LSB")
    return rows


if __name__ == "__main__":
    W, H = int(os.environ.get("W", "512")), int(os.environ.get("H", "288"))
    scales = tuple(int(x) for x in os.environ.get("SCALES", "2,3,4").split(","))
    rows = sweep(W, H, scales)
    worst = {p: max(r[p]["max_lsb"] for r in rows) for p in ("direct", "winograd", "auto")}
    first2 = {p: next((f"{r['draw']} x{r['scale']} {r['frame']}" for r in rows if r[p]["max_lsb"] >= 2), None) for p in ("direct", "winograd", "auto")}
    summary = {"frame": [W, H], "draws": len(synth.WEIGHT_DRAWS), "cases": len(rows), "worst_max_lsb": worst, "first_case_with_2_lsb": first2,
               "worst_fraction_differing": {p: max(r[p]["fraction_differing"] for r in rows) for p in ("direct", "winograd", "auto")},
               # the rule's verdict: where auto chose Winograd it must be within 1 LSB; where it refused, Winograd must indeed be worse than 1 LSB somewhere
               "auto_chose_winograd_draws": sorted({r["draw"] for r in rows if r["auto_chose_winograd"]}),
               "auto_kept_direct_draws": sorted({r["draw"] for r in rows if not r["auto_chose_winograd"]}),
               "worst_max_lsb_where_auto_chose_winograd": max((r["winograd"]["max_lsb"] for r in rows if r["auto_chose_winograd"]), default=0),
               "kappa_range_chosen": [min(r["kappa"] for r in rows if r["auto_chose_winograd"]), max(r["kappa"] for r in rows if r["auto_chose_winograd"])],
               "kappa_range_refused": [min((r["kappa"] for r in rows if not r["auto_chose_winograd"]), default=None), max((r["kappa"] for r in rows if not r["auto_chose_winograd"]), default=None)]}
    print(json.dumps(summary))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"summary": summary, "cases": rows}, open("gpurun_out/parity_sweep.json", "w"), indent=1)
