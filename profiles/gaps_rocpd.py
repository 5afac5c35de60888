"""GPU idle time between kernels from a rocprofv3 rocpd database: busy fraction of the last `frac` of the trace and the idle time
attributed to the kernel that PRECEDES each gap (a host round trip after that kernel shows up here).
Usage: python profiles/gaps_rocpd.py DB [frac=0.5] [min_gap_us=3]"""
import sqlite3
import sys
from collections import defaultdict


def main(db, frac=0.5, min_gap_us=3.0):
    c = sqlite3.connect(db)
    rows = c.execute("""select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                        on d.kernel_id = s.id order by d.start""").fetchall()
    rows = rows[int(len(rows) * (1.0 - float(frac))):]
    span = rows[-1][2] - rows[0][1]
    busy = sum(e - s for _, s, e in rows)
    gaps = defaultdict(lambda: [0, 0.0])
    for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
        g = (s1 - e0) / 1e3
        if g >= float(min_gap_us):
            short = n0.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            gaps[short][0] += 1
            gaps[short][1] += g
    print(f"kernels {len(rows)}  span_ms {span / 1e6:.2f}  busy_ms {busy / 1e6:.2f}  busy_frac {busy / span:.3f}")
    for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  gap after {k:<28} x{n:<6} total_ms {t / 1e3:8.2f}  avg_us {t / n:7.1f}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
