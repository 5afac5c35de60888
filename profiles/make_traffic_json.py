"""profiles/r02_traffic.json from the two rocprofv3 PMC passes of the bench command (profiles/run_r04_profiles.sh):
FETCH_SIZE and WRITE_SIZE are reported in KiB per dispatch; FETCH_SIZE is doubled for gfx950 as MI355X_MICROARCH.md prescribes
(the counter tallies 128-byte requests at 64 bytes), WRITE_SIZE is taken as is (uncalibrated there).
    python profiles/make_traffic_json.py profiles/r02_a_bench_pmc_fetch.csv profiles/r02_a_bench_pmc_write.csv profiles/r02_traffic.json"""
import json
import sys


def read(path, counter):
    out = {}
    for line in open(path):
        parts = line.rstrip("\n").rsplit(",", 3)
        if len(parts) != 4 or parts[1] != counter:
            continue
        name, _, n, avg = parts
        out[name] = (int(n), float(avg))
    return out


def main(fetch_csv, write_csv, out_json):
    f, w = read(fetch_csv, "FETCH_SIZE"), read(write_csv, "WRITE_SIZE")
    kernels = {}
    for name, (n, kib) in f.items():
        short = name.split("<")[0]
        row = {"symbol": name, "launches": n, "fetch_raw_kb": kib, "fetch_bytes": 2.0 * kib * 1024.0,
               "write_bytes": w.get(name, (0, 0.0))[1] * 1024.0}
        kernels.setdefault(short, []).append(row)
    kernels = {k: (v[0] if len(v) == 1 else v) for k, v in kernels.items()}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --no-micro "
                         "--steps 10 --warmup 3; " + fetch_csv + ", " + write_csv,
               "corrections": "FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 bytes), KiB -> bytes; WRITE_SIZE KiB -> bytes as reported",
               "kernels": kernels}, open(out_json, "w"), indent=1)
    print("kernels:", len(kernels))


if __name__ == "__main__":
    main(*sys.argv[1:4])
