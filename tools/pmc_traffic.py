"""HBM-side traffic per launch of one kernel family from two rocprofv3 PMC passes of the same command.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -o fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT -o write -- python3 bench.py ... (same)
    python tools/pmc_traffic.py OUT/fetch_counter_collection.csv OUT/write_counter_collection.csv --kernel sg_igemm_kernel --out profiles/r01_igemm_traffic.json

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests of 16-byte-per-lane loads at 64 bytes, so it is doubled;
WRITE_SIZE is exact for 16-byte stores and float atomics.  (The separate passes are required: the two counters do
not fit the TCC slots together.)"""
import argparse
import csv
import json
import sys

csv.field_size_limit(sys.maxsize)


def collect(path, counter, kernel):
    total, launches = 0.0, 0
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter or kernel not in row["Kernel_Name"]:
                continue
            total += float(row["Counter_Value"])
            launches += 1
    return total, launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--kernel", default="sg_igemm_kernel")
    ap.add_argument("--out", default=None)
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    f_kib, nf = collect(a.fetch_csv, "FETCH_SIZE", a.kernel)
    w_kib, nw = collect(a.write_csv, "WRITE_SIZE", a.kernel)
    if nf == 0 or nw == 0:
        raise SystemExit("no %s dispatches with FETCH_SIZE / WRITE_SIZE found" % a.kernel)
    fetch_b = 2.0 * f_kib * 1024.0 / nf        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
    write_b = w_kib * 1024.0 / nw
    res = {"kernel": a.kernel, "launches_fetch_pass": nf, "launches_write_pass": nw,
           "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
           "traffic_bytes_per_launch": fetch_b + write_b,
           "correction": "FETCH_SIZE[KiB] x 1024 x 2 (gfx950 half-count of 16-B/lane loads), WRITE_SIZE[KiB] x 1024",
           "command": a.command}
    s = json.dumps(res, indent=1)
    print(s)
    if a.out:
        open(a.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
