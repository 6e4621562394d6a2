#!/usr/bin/env python3
"""Turn rocprofv3 output directories (scratch, under gpurun_out/) into the committed summaries under profiles/.

    python tools/summarize_profile.py --round 1 --stats gpurun_out/prof_stats --fetch gpurun_out/prof_fetch \
        --write gpurun_out/prof_write

* ``profiles/rNN_kernel_stats.csv``   -- rocprofv3 --kernel-trace --stats summary (verbatim *_kernel_stats.csv)
* ``profiles/rNN_pmc_traffic.md``     -- per-kernel HBM traffic from the FETCH_SIZE / WRITE_SIZE passes
* ``profiles/pmc_traffic.json``       -- the same numbers keyed by kernel name, read by bench.py's ``roofline.traffic``

Counter handling follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are collected in separate passes
(they do not fit one pass), both are in KiB, and on gfx950 FETCH_SIZE reports exactly half the bytes of a wide
(16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
from __future__ import annotations

import argparse
import collections
import csv
import json
import re
import shutil
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def find(dirpath: Path, suffix: str) -> Path:
    hits = sorted(dirpath.rglob(f"*{suffix}"))
    if not hits:
        raise SystemExit(f"no *{suffix} under {dirpath}")
    return hits[-1]


def short(name: str) -> str:
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def counter_means(path: Path, counter: str) -> dict[str, tuple[float, int]]:
    acc = collections.defaultdict(list)
    with path.open() as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, required=True)
    ap.add_argument("--stats", type=Path)
    ap.add_argument("--fetch", type=Path)
    ap.add_argument("--write", type=Path)
    ap.add_argument("--command", default="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline")
    ap.add_argument("--name", default="pmc_traffic", help="profiles/rNN_<name>.md (pmc_traffic.json only for the default)")
    ap.add_argument("--alg-bytes", type=float, default=0.0, help="algorithmic bytes per launch, for a ratio column")
    args = ap.parse_args()
    out = REPO / "profiles"
    out.mkdir(exist_ok=True)
    tag = f"r{args.round:02d}"
    if args.stats:
        src = find(args.stats, "_kernel_stats.csv")
        dst = out / f"{tag}_kernel_stats.csv"
        shutil.copyfile(src, dst)
        print(f"wrote {dst}")
    if args.fetch and args.write:
        fetch = counter_means(find(args.fetch, "_counter_collection.csv"), "FETCH_SIZE")
        write = counter_means(find(args.write, "_counter_collection.csv"), "WRITE_SIZE")
        table = {}
        lines = [f"# {tag}: HBM traffic per launch from rocprofv3 PMC passes",
                 "",
                 f"Command (each pass): `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- {args.command}`",
                 "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as read; both KiB -> bytes.",
                 "",
                 "| kernel | launches | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes / launch | GiB | read GiB | written GiB"
                 + (" | read / algorithmic | written / algorithmic |" if args.alg_bytes else " |"),
                 "|---|---|---|---|---|---|---|---|" + ("---|---|" if args.alg_bytes else "")]
        for k in sorted(set(fetch) & set(write)):
            f, nf = fetch[k]
            w, _ = write[k]
            hbm = (2.0 * f + w) * 1024.0
            table[k] = {"hbm_bytes_per_launch": hbm, "fetch_kib_raw": f, "write_kib": w, "launches": nf,
                        "source": f"profiles/{tag}_pmc_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                  "separate passes, FETCH_SIZE x2)"}
            row = (f"| `{k}` | {nf} | {f:.1f} | {w:.1f} | {hbm:.4g} | {hbm / 2**30:.3f} | {2 * f / 2**20:.3f} | "
                   f"{w / 2**20:.3f} |")
            if args.alg_bytes:
                row += f" {2 * f * 1024 / (args.alg_bytes / 2):.3f} | {w * 1024 / (args.alg_bytes / 2):.3f} |"
            lines.append(row)
        (out / f"{tag}_{args.name}.md").write_text("\n".join(lines) + "\n")
        if args.name == "pmc_traffic":
            (out / "pmc_traffic.json").write_text(json.dumps(table, indent=1, sort_keys=True) + "\n")
        print(f"wrote {out / (tag + '_' + args.name + '.md')}")


if __name__ == "__main__":
    main()
