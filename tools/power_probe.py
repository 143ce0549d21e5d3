"""Clock / power while bench-like frames render back to back: python tools/power_probe.py [--overlap] [--only features|mlp]"""
import argparse, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--overlap", action="store_true")
ap.add_argument("--seconds", type=float, default=5.0)
a = ap.parse_args()
import torch, bench
dev = torch.device("cuda", 0)
model, cfg, sd = bench.build_model(dev)
model.overlap_streams = a.overlap
batch = bench.frame_rays(dev)
flat = {k: v.reshape(-1, v.shape[-1]) for k, v in batch.items()}
samples, stop = [], False
def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        sclk = [l.split("(")[1].split(")")[0] for l in out.splitlines() if "sclk" in l]
        pw = [l.split(":")[-1].strip() for l in out.splitlines() if "Power (W)" in l]
        samples.append((sclk[0] if sclk else "?", pw[0] if pw else "?"))
        time.sleep(0.25)
t = threading.Thread(target=poll); t.start()
t0 = time.time(); k = 0
with torch.no_grad():
    while time.time() - t0 < a.seconds:
        model._march(False, flat, 1.0, True, None, want_history=False)
        torch.cuda.synchronize(); k += 1
dt = time.time() - t0
stop = True; t.join()
print(f"overlap={a.overlap}: {dt / k * 1e3:.1f} ms per frame; (sclk, W) samples: {samples[3:15]}")
