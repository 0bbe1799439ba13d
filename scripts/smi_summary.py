"""Summarise rocm-smi --showclocks --showpower --json samples (one JSON object per line) taken while a profiling pass ran."""
import json, re, sys
sclk, power = [], []
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith("{"):
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    for card, f in d.items():
        if not isinstance(f, dict):
            continue
        for k, v in f.items():
            m = re.search(r"([0-9.]+)", str(v))
            if not m:
                continue
            if "sclk" in k.lower() and "level" not in k.lower() or k.lower().startswith("sclk clock speed"):
                sclk.append(float(m.group(1)))
            elif "power" in k.lower() and "socket" in k.lower():
                power.append(float(m.group(1)))
        break
def stat(a):
    a = sorted(a)
    return "n=%d min %.0f median %.0f max %.0f" % (len(a), a[0], a[len(a) // 2], a[-1]) if a else "no samples"
print("\n## rocm-smi during this pass (1 Hz; the pass is mostly host set-up -- the busy samples are the high-power ones)\n")
print("* sclk [MHz]: %s" % stat(sclk))
print("* socket power [W]: %s" % stat(power))
busy = [s for s, p in zip(sclk, power) if power and p >= 0.6 * max(power)] if len(sclk) == len(power) else []
print("* sclk of the samples at >= 60 %% of the peak power: %s" % stat(busy))
