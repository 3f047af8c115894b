import os, sys, json, subprocess
for g in (1, 4, 8):
    env = dict(os.environ, BWAHIP_SMEM_LANES=str(g))
    r = subprocess.run([sys.executable, "bench.py", "--genome-mbp", "128", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(g, d["kernel_ms"]["k_smem"], d["per_read"], flush=True)
