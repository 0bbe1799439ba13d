"""Clock and socket power the part sustains under each conv family (rocm-smi sampled while one layer's kernel runs back to back for a few
seconds): the number every 'fraction of peak' in DESIGN.md has to be read against.   python scripts/smi_during_kernels.py"""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dpig_amd import hip_ops as H
dev = torch.device("cuda:0")


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(txt[txt.index("{"):])
            f = next(v for v in d.values() if isinstance(v, dict))
            s = p = None
            for k, v in f.items():
                m = re.search(r"([0-9.]+)", str(v))
                if not m:
                    continue
                if "sclk" in k.lower() and "level" not in k.lower():
                    s = float(m.group(1))
                elif "power" in k.lower() and "socket" in k.lower():
                    p = float(m.group(1))
            out.append((s, p))
        except Exception:
            pass


def run(name, fn, flops, secs=4.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    t0 = time.time()
    th.start()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    dt = e0.elapsed_time(e1) * 1e-3
    out = out[1:] if len(out) > 2 else out             # (the first sample may predate the ramp)
    sc = sorted(s for s, _ in out if s)
    pw = sorted(p for _, p in out if p)
    med = lambda a: a[len(a) // 2] if a else float("nan")
    print("%-44s %7.1f TFLOP/s executed | sclk median %5.0f MHz (min %5.0f) | socket power median %4.0f W (max %4.0f) | %d samples" % (
        name, flops * n / dt / 1e12, med(sc), sc[0] if sc else float("nan"), med(pw), pw[-1] if pw else float("nan"), len(out)))


g = torch.Generator(device=dev).manual_seed(0)
N, Hh, W, C = 16, 64, 32, 512
x = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
w = (torch.rand((3, 3, C, C), device=dev, generator=g) * 2 - 1) * 0.02
b = torch.rand((C,), device=dev, generator=g)
dy = torch.rand((N, Hh, W, C), device=dev, generator=g) * 2 - 1
dw = torch.empty_like(w)
direct = 2.0 * N * Hh * W * 9 * C * C
H.set_compute("f32")
run("direct fp32 conv fwd, dec3 64x32 C512", lambda: H.conv2d_fwd(x, w, b, act=1), direct)
H.set_compute("f32w"); H.set_wino_mode(2)
w._dpig_wino = H.wino_images(w)
run("Winograd fwd, dec3 (executed = 16/36)", lambda: H.conv2d_fwd(x, w, b, act=1), direct * 16 / 36)
run("Winograd filter gradient, dec3", lambda: H.conv2d_wgrad(x, dy, (3, 3, C, C), out=dw), direct * 16 / 36)
H.set_wino_mode(1)
H.set_compute("bf16")
xb, dyb = x.to(torch.bfloat16), dy.to(torch.bfloat16)
run("bf16 conv fwd, dec3", lambda: H.conv2d_fwd(xb, w, b, act=1), direct)
H.set_compute("f32")
time.sleep(1.0)
stop, out = threading.Event(), []
th = threading.Thread(target=sample, args=(stop, out)); th.start(); time.sleep(2.0); stop.set(); th.join()
print("idle: %s" % (out[-1],))
