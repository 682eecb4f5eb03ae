#!/usr/bin/env python3
"""A MEMO index from SEQUENCES instead of the uniform generator (round-2 VERDICT item 6).

    python tools/realistic_index.py --length 5000000 --genomes 50 --out gpurun_out/real

1. a random pivot of --length bases and --genomes - 1 mutated copies: SNPs at a per-genome rate drawn from
   [--snp-lo, --snp-hi] (0.1 % .. 1 %), short indels at a tenth of that rate, a few inversions and translocated
   segments, one deletion of a long stretch in a fifth of the genomes;
2. matching statistics of the pivot against every genome + its reverse complement (the text index.sh:63-65 builds),
   by suffix automaton on the host cores (tools/ms_sam.cpp, compiled here with g++);
3. the DAP matrix -> index rows on the GPU (memo_amd.dap_to_bed.DapConverter = dap_to_bed.py:55-134 with --mem --overlap,
   --order for the conservation index) -> `cons.npz` / `memb.npz` (start, end, annot, num_docs) and `cons.parquet`;
4. statistics of the rows: rows per position, overlap lengths, the share of rows that cannot write at k = 21 / 31 / 101.

GPU box only for step 3 (the library has no CPU fallback); steps 1-2 run anywhere.  Development tool."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def revcomp(x):
    out = x[::-1].copy()
    m = out < 4
    out[m] = 3 - out[m]
    return out


def mutate(rng, pivot, snp, genome_no):
    """a copy of the pivot with SNPs, short indels, a few rearrangements (uint8 codes 0..3)"""
    L = len(pivot)
    g = pivot.copy()
    hit = rng.random(L) < snp
    g[hit] = (g[hit] + rng.integers(1, 4, int(hit.sum()))) & 3                 # substitutions
    keep = rng.random(L) >= snp / 20                                          # 1-base deletions
    ins = np.flatnonzero(rng.random(L) < snp / 20)                            # short insertions (1-6 bases)
    pieces, at = [], 0
    g = g[keep]
    ins = ins[ins < len(g)]
    for p in ins:
        pieces.append(g[at:p])
        pieces.append(rng.integers(0, 4, int(rng.integers(1, 7))).astype(np.uint8))
        at = p
    pieces.append(g[at:])
    g = np.concatenate(pieces)
    for _ in range(int(rng.integers(0, 4))):                                  # inversions of 1-50 kb
        a = int(rng.integers(0, len(g) - 60_000))
        n = int(rng.integers(1_000, 50_000))
        g[a:a + n] = revcomp(g[a:a + n])
    for _ in range(int(rng.integers(0, 3))):                                  # translocations of 5-100 kb
        a = int(rng.integers(0, len(g) - 120_000))
        n = int(rng.integers(5_000, 100_000))
        seg = g[a:a + n].copy()
        g = np.concatenate([g[:a], g[a + n:]])
        b = int(rng.integers(0, len(g)))
        g = np.concatenate([g[:b], seg, g[b:]])
    if genome_no % 5 == 0:                                                    # a long stretch this genome lacks
        a = int(rng.integers(0, len(g) - L // 20))
        g = np.concatenate([g[:a], g[a + L // 50:]])
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--genomes", type=int, default=50, help="genomes in the pangenome, pivot included")
    ap.add_argument("--snp-lo", type=float, default=0.001)
    ap.add_argument("--snp-hi", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=20260)
    ap.add_argument("--out", default="gpurun_out/real")
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 8, 32))
    ap.add_argument("--chunks", type=int, default=1,
                    help="C independent pangenomes of --length bases each (seeds --seed, --seed + 1, ...), laid end to end on one pivot of "
                         "C x length positions: an index too big for the 256 MiB Infinity Cache without a suffix automaton of C x length "
                         "bases (round 4: 8 x 20 Mbp x 50 genomes = 7.5e8 rows)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.chunks > 1:
        return chunked(a)
    one_pangenome(a, a.seed, a.out, True)


def chunked(a):
    import copy
    cols = {"cons": [[], [], []], "memb": [[], [], []]}
    per_chunk = []
    for c in range(a.chunks):
        sub = os.path.join(a.out, "chunk")
        os.makedirs(sub, exist_ok=True)
        st = one_pangenome(copy.copy(a), a.seed + c, sub, False)
        per_chunk.append({"seed": a.seed + c, "seconds": st["seconds"], "cons_rows": st["cons"]["rows"], "memb_rows": st["memb"]["rows"]})
        for name in ("cons", "memb"):
            z = np.load(os.path.join(sub, name + ".npz"))
            cols[name][0].append(z["start"] + c * a.length)
            cols[name][1].append(z["end"] + c * a.length)
            cols[name][2].append(z["annot"])
            os.unlink(os.path.join(sub, name + ".npz"))
    L = a.chunks * a.length
    stats = {"length": L, "genomes": a.genomes, "chunks": a.chunks, "chunk_length": a.length, "per_chunk": per_chunk}
    for name in ("cons", "memb"):
        start, end, annot = (np.concatenate(x) for x in cols[name])
        cols[name] = None
        assert bool(np.all(np.diff(start) >= 0))
        np.savez(os.path.join(a.out, name + ".npz"), start=start, end=end, annot=annot, num_docs=a.genomes, length=L)
        stats[name] = row_stats(start, end, annot, L)
    with open(os.path.join(a.out, "index_stats.json"), "w") as fh:
        json.dump(stats, fh, indent=1)
    print(json.dumps(stats))


def row_stats(start, end, annot, L):
    rows_at = np.bincount(start, minlength=L + 1)
    ov = end - start
    return {"rows": int(len(start)), "rows_per_position": float(len(start) / L),
            "start_sorted": bool(np.all(np.diff(start) >= 0)), "max_annot": int(annot.max()),
            "positions_with_rows": float((rows_at > 0).mean()),
            "rows_per_position_histogram": {str(q): int(np.quantile(rows_at[::7], q)) for q in (0.5, 0.9, 0.99, 0.999, 1.0)},
            "overlap_histogram": {f"{lo}-{hi}": float(((ov >= lo) & (ov < hi)).mean())
                                  for lo, hi in ((0, 1), (1, 8), (8, 16), (16, 20), (20, 30), (30, 63), (63, 100), (100, 256), (256, 1 << 40))},
            "rows_that_cannot_write": {str(k): float((ov >= k - 1).mean()) for k in (21, 31, 64, 101)}}


def one_pangenome(a, seed, out, final):
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    L, N = a.length, a.genomes
    a.out = out
    pivot = rng.integers(0, 4, L).astype(np.uint8)
    pivot.tofile(os.path.join(a.out, "pivot.bin"))
    rates, paths = [], []
    for g in range(1, N):
        snp = float(np.exp(rng.uniform(np.log(a.snp_lo), np.log(a.snp_hi))))
        rates.append(snp)
        seq = mutate(rng, pivot, snp, g)
        text = np.concatenate([seq, np.array([4], np.uint8), revcomp(seq), np.array([4], np.uint8)])
        p = os.path.join(a.out, f"g{g}.bin")
        text.tofile(p)
        paths.append(p)
    t1 = time.perf_counter()
    exe = os.path.join(a.out, "ms_sam")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tools", "ms_sam.cpp"), "-o", exe])
    dap_path = os.path.join(a.out, "dap.i32")
    subprocess.check_call([exe, os.path.join(a.out, "pivot.bin"), dap_path] + paths, env=dict(os.environ, MS_THREADS=str(a.threads)))
    for p in paths:
        os.unlink(p)
    t2 = time.perf_counter()
    dap = np.fromfile(dap_path, np.int32).reshape(L, N - 1)
    os.unlink(dap_path)

    from memo_amd.dap_to_bed import DapConverter
    rec_begin = np.array([0, L], np.int64)
    stats = {"length": L, "genomes": N, "snp_rates": [round(r, 5) for r in rates], "seconds": {"genomes": t1 - t0, "matching_statistics": t2 - t1},
             "ms_mean": float(dap.mean()), "ms_median": float(np.median(dap[::97]))}
    for name, order in (("cons", True), ("memb", False)):
        tq = time.perf_counter()
        parts = []
        with DapConverter(N - 1, rec_begin, order, True) as conv:
            for b in range(0, L, 1 << 20):
                parts.append(conv.push(dap[b:b + (1 << 20)]))
            parts.append(conv.finish())
        start = np.concatenate([p[1] for p in parts])
        end = np.concatenate([p[2] for p in parts])
        annot = np.concatenate([p[3] for p in parts]).astype(np.int64)
        np.savez(os.path.join(a.out, name + ".npz"), start=start, end=end, annot=annot, num_docs=N, length=L)
        rows_at = np.bincount(start, minlength=L + 1)
        ov = end - start
        st = {"rows": int(len(start)), "rows_per_position": float(len(start) / L), "dap_to_rows_s": time.perf_counter() - tq,
              "start_sorted": bool(np.all(np.diff(start) >= 0)), "max_annot": int(annot.max()),
              "positions_with_rows": float((rows_at > 0).mean()),
              "rows_per_position_histogram": {str(q): int(np.quantile(rows_at, q)) for q in (0.5, 0.9, 0.99, 0.999, 1.0)},
              "overlap_histogram": {f"{lo}-{hi}": float(((ov >= lo) & (ov < hi)).mean())
                                    for lo, hi in ((0, 1), (1, 8), (8, 16), (16, 20), (20, 30), (30, 63), (63, 100), (100, 256), (256, 1 << 40))},
              "rows_that_cannot_write": {str(k): float((ov >= k - 1).mean()) for k in (21, 31, 64, 101)}}
        stats[name] = st
        if name == "cons" and final:
            import pyarrow as pa
            import pyarrow.parquet as pq
            pq.write_table(pa.table({"f0": pa.array(["chr1"] * len(start), pa.utf8()), "f1": start, "f2": end, "f3": annot}),
                           os.path.join(a.out, "cons.parquet"), compression="ZSTD", row_group_size=1 << 20)
    if final:
        with open(os.path.join(a.out, "index_stats.json"), "w") as fh:
            json.dump(stats, fh, indent=1)
        print(json.dumps(stats))
    return stats


if __name__ == "__main__":
    main()
