#!/usr/bin/env python3
"""End-to-end `memo query` timing on a synthetic Parquet index (GPU box).
  python tools/cli_timing.py --num-docs 100 --pivot 20000000 --out gpurun_out/cli
  python tools/cli_timing.py --num-docs 100 --pivot 100000000 --memb-window 10000000     # BASELINE config 3 itself: 5e8 rows of Parquet"""
import argparse
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memo_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--num-docs", type=int, default=100)
ap.add_argument("--pivot", type=int, default=20_000_000)
ap.add_argument("--out", default="/tmp/cli_timing")
ap.add_argument("--memb-window", type=int, default=2_000_000, help="positions of the membership query's window")
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
pq_path = os.path.join(a.out, f"synth_n{a.num_docs}_{a.pivot}.parquet")
t = time.time()
rows = synth.write_parquet(pq_path, a.num_docs, a.pivot)
print(f"wrote {rows} rows, {os.path.getsize(pq_path) / 1e6:.0f} MB parquet in {time.time() - t:.1f} s", flush=True)
exe = os.path.join(ROOT, "bin", "memo")
import shutil
shutil.rmtree(pq_path + ".memo", ignore_errors=True)
# three passes over the same queries: no sidecar cache at all; a miss that builds the cache in-process after the
# answer is written (MEMO_CACHE=sync: its wall clock includes the build; the default builds in a detached process);
# hits.  Output bytes must not change.
digests = {}
for mode, what in (("0", "no cache (Parquet decode)"), ("sync", "cache miss + in-process build of the cache"), ("read", "cache hit"),
                   ("read", "cache hit, again")):
    print(f"-- MEMO_CACHE={mode}: {what}")
    for region, memb in ((f"chr1:0-{a.pivot}", False), (f"chr1:{a.pivot // 4}-{a.pivot // 2}", False),
                         (f"chr1:0-{min(a.pivot, a.memb_window)}", True)):
        out = os.path.join(a.out, "out.txt")
        argv = [sys.executable, exe, "query", "-b", pq_path, "-n", str(a.num_docs), "-r", region, "-o", out] + (["-m"] if memb else [])
        t = time.time()
        r = subprocess.run(argv, capture_output=True, env=dict(os.environ, MEMO_TIMING="1", MEMO_CACHE=mode))
        wall = time.time() - t
        sz = os.path.getsize(out) if os.path.exists(out) else -1
        h = hashlib.sha256(open(out, "rb").read()).hexdigest()[:16] if sz >= 0 else None
        same = digests.setdefault((region, memb), h) == h
        print(f"{region} {'membership' if memb else 'conservation'}: wall {wall:.2f} s, rc {r.returncode}, output {sz / 1e6:.0f} MB, "
              f"sha256 {h}{'' if same else '  <-- DIFFERS'}")
        lines = [ln for ln in r.stderr.decode().strip().splitlines() if "memo_query timing" in ln]
        print("   ", lines[-1] if lines else r.stderr.decode()[-300:])
cache_dir = pq_path + ".memo"
if os.path.isdir(cache_dir):
    print("cache files:", {f: os.path.getsize(os.path.join(cache_dir, f)) for f in os.listdir(cache_dir)})
