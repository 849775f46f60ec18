"""Samples the GPU's shader clock and socket power from sysfs (hwmon) while a child command runs:
    python tools/exp/clock_watch.py python bench.py --no-extras --cpu-cubes 0 --steps 40
This process never touches the GPU; the child does."""
import glob
import subprocess
import sys
import time


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def main():
    hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    print("hwmon dirs:", hw)
    files = {}
    for d in hw:
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input"):
            if read(d + "/" + name) is not None:
                files[d.split("/")[4] + ":" + name] = d + "/" + name
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("pp_dpm_sclk", "gpu_busy_percent"):
            if read(d + "/" + name) is not None:
                files[d.split("/")[4] + ":" + name] = d + "/" + name
    print("sampling:", sorted(files))
    import os
    probe = subprocess.run([sys.executable, "-c", "import torch; p = torch.cuda.get_device_properties(0); "
                            "print('%04x:%02x:%02x' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().strip()
    print("device 0 of the child is at PCI", probe, "; visible:", os.environ.get("ROCR_VISIBLE_DEVICES"), os.environ.get("HIP_VISIBLE_DEVICES"))
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        print("  ", d.split("/")[4], os.path.basename(os.path.realpath(d)))
    child = subprocess.Popen(sys.argv[1:], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    rows = []
    t0 = time.time()
    while child.poll() is None:
        rows.append((time.time() - t0, {k: read(p) for k, p in files.items()}))
        time.sleep(0.05)
    out = child.stdout.read().decode()
    print(out[-400:])
    for k in sorted(files):
        if k.endswith("pp_dpm_sclk"):
            cur = [[ln for ln in (r[1][k] or "").splitlines() if "*" in ln] for r in rows]
            vals = [c[0] for c in cur if c]
            print(k, "first", vals[:1], "distinct", sorted(set(vals))[:12])
            continue
        v = []
        for r in rows:
            try:
                v.append(float(r[1][k]))
            except (TypeError, ValueError):
                pass
        if v:
            v2 = sorted(v)
            print("%-28s n=%d min %.4g p10 %.4g median %.4g p90 %.4g max %.4g" % (k, len(v), v2[0], v2[len(v2) // 10], v2[len(v2) // 2], v2[9 * len(v2) // 10], v2[-1]))
    # time series (1 sample per 0.5 s) of the first freq/power files
    mine = [os.path.basename(os.path.dirname(d)) for d in glob.glob("/sys/class/drm/card*/device")
            if os.path.basename(os.path.realpath(d)).lower().startswith(probe.lower()[:10])]
    print("my card:", mine)
    keys = [k for k in sorted(files) if k.split(":")[0] in mine and k.endswith(("freq1_input", "power1_input", "gpu_busy_percent"))]
    for t, r in rows[::5]:
        print("%6.2f " % t + "  ".join("%s=%s" % (k.split(":")[1], r[k]) for k in keys))
    return child.returncode


if __name__ == "__main__":
    sys.exit(main())
