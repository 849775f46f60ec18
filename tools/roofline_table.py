"""Markdown table of every conv kernel of one bench step from a bench.py JSON line produced with PCGC_BENCH_TOP=40:
    PCGC_BENCH_TOP=40 python bench.py --no-extras --cpu-cubes 0 --steps 10 > b.json ; python tools/roofline_table.py b.json"""
import json
import sys


def main(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    r = d["roofline"]
    peak = r["peak"]
    clock = r.get("clock") or {}
    out = ["# Per-kernel table of one bench step (%s cubes, encode + decode)" % d["config"].get("cubes", "205"), "",
           "`PCGC_BENCH_TOP=40 python bench.py --no-extras --cpu-cubes 0 --steps %d` on one MI355X: %.0f cubes/s, %.2f ms per step; per-launch"
           % (d["steps"], d["value"], d["ms_per_step"]),
           "hipEvent times of two single-pipeline steps (`roofline` block of bench.py).  fp32 MFMA peak %.1f TFLOP/s; HBM-bound" % peak,
           "layers are marked with their algorithmic GB/s instead.  All conv kernels together: %.1f TFLOP/s, %.2f ms per step."
           % (r["all_conv_tflops"], r["conv_ms_per_step"])]
    if clock:
        out += ["Shader clock during the timed steps: median %d MHz (%d-%d), socket power %s W: the peak at that clock is %.1f TFLOP/s;"
                % (clock["sclk_mhz_median"], clock["sclk_mhz_min"], clock["sclk_mhz_max"], clock.get("socket_power_w_median", "?"),
                   clock["peak_at_measured_clock"]),
                "the last column is the fraction of THAT peak."]
    out += ["The memory column is FETCH_SIZE x 2 + WRITE_SIZE per launch (committed `--pmc` passes) over the launch's live duration,",
            "as a fraction of the 8 TB/s HBM peak: `vrn16bc_row_kernel` sits at the memory roofline, not at the matrix one."]
    out += ["", "| kernel | ms per step | share of conv time | TFLOP/s | fraction of %.1f | at the measured clock | through L2, TB/s (of 8) |" % peak, "|---|---|---|---|---|---|---|"]
    tot = r["conv_ms_per_step"]
    pc = clock.get("peak_at_measured_clock")
    for k in r["top_kernels"]:
        if k.get("bound") == "hbm":
            frac, fc = "HBM-bound: %.0f GB/s algorithmic" % k["algorithmic_GBps"], ""
        else:
            frac, fc = "%.2f" % (k["tflops"] / peak), ("%.2f" % (k["tflops"] / pc)) if pc else ""
        mem = ("%.1f (%.2f)" % (k["hbm_GBps"] / 1e3, k["hbm_frac"])) if "hbm_GBps" in k else ""
        out.append("| `%s` | %.3f | %.1f %% | %.1f | %s | %s | %s |" % (k["kernel"], k["ms_per_step"], 100 * k["ms_per_step"] / tot, k["tflops"], frac, fc, mem))
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
